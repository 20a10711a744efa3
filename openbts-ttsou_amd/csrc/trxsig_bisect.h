// trxsig_bisect.h -- peakDetect's early/late bisection (sigProcLib.cpp:684-701) in its two device forms:
// peak_bisect (one burst per lane, the reference's serial loop) and the speculative form (fused_point /
// fused_decide + FusedRel tables: the lanes of a burst evaluate every node of the next levels at once and the
// decisions are replayed along the path the reference takes), plus analyzeTrafficBurst's tail (fused_tail).
// Numerical contract: see trxsig_dev.h / DESIGN.md (every float32 operation is the reference's, in the
// reference's order; built with -ffp-contract=off).
#pragma once
#include "trxsig_dev.h"

namespace {

// ---------------------------------------------------------------------------------------------
// peak_bisect: peakDetect's early-late bisection (sigProcLib.cpp:684-701) for ONE burst per lane.
//   loc[j][lane], j = 0..23, holds the correlation at lags M-12 .. M+11 (M = integer argmax) with
//   zeros wherever interpolatePoint would skip a term (lag < 0, lag > n-2); loc[24], loc[25] = 0.
//
// interpolatePoint(ix) = sum_{i} corr[i]*sinc(pi*(i-ix)), i from max(0,floor(ix)-10) to
// min(floor(ix)+11, n-1)-1.  ix stays on the 1/512 grid, so sinc(pi*(i-ix)) comes from
// sinc_grid[f][j] with f = frac(ix)*512, j = i-floor(ix)+10 (trxsig_tables.h).  early and late
// differ by exactly 2.0, hence share f.  The row needed by the next step is known one step ahead up
// to the sign of the early/late decision, so both candidates are fetched while the current step
// computes.  Returns the interpolated peak; *peakIx = its (fractional) index.
// ---------------------------------------------------------------------------------------------
template <int LW, typename TAB>
__device__ __forceinline__ cx peak_bisect(const TAB &tab, const cx (*loc)[LW], int lane, int M, float *peakIx) {
  // tab: float[512][24], either TrxTables::sinc_grid in global memory or a copy in LDS
  // Packed float32 pairs (trxsig_dev.h): a row is kept as 12 register pairs of two consecutive taps, and a complex-by-real
  // multiply-add is one v_pk_mul_f32 (tap picked by op_sel) + one v_pk_add_f32 -- the same products and sums, each rounded on
  // its own, in half the instructions; the lane-per-burst kernels that call this are bound by what one wave can issue.
  auto load_row = [&](int f, v2f (&s)[12]) {
    const float4 *row = reinterpret_cast<const float4 *>(tab[f & 511]);
#pragma unroll
    for (int q = 0; q < 6; q++) {
      const float4 v = row[q];
      s[2 * q].x = v.x; s[2 * q].y = v.y; s[2 * q + 1].x = v.z; s[2 * q + 1].y = v.w;
    }
  };
  // pa = interpolatePoint(ix), pb = interpolatePoint(ix + dI2): same fractional part, row s
  auto interp2 = [&](float ix, int dI2, const v2f (&s)[12], v2f &pa, v2f &pb) {
    const int I = (int)floorf(ix);
    int base = I - M + 2;                                  // loc index of tap j = 0 (0..3 by construction)
    base = base < 0 ? 0 : (base > 3 ? 3 : base);
    pa = pk(mk(0, 0)); pb = pk(mk(0, 0));
#pragma unroll
    for (int j = 0; j < 21; j++) {
      const v2f xa = pk(loc[base + j][lane]), xb = pk(loc[base + j + dI2][lane]);
      if (j & 1) { pa = pk_cadd(pa, pk_mul_tap<1>(xa, s[j >> 1])); pb = pk_cadd(pb, pk_mul_tap<1>(xb, s[j >> 1])); }
      else { pa = pk_cadd(pa, pk_mul_tap<0>(xa, s[j >> 1])); pb = pk_cadd(pb, pk_mul_tap<0>(xb, s[j >> 1])); }
    }
  };
  auto frac512 = [](float ix) { return (int)((ix - floorf(ix)) * 512.0f); };

  float early = (float)M - 1;
  float incr = 0.5f;
  bool active = true;
  v2f cur[12], up[12], dn[12];
  load_row(0, cur);                                        // early = M-1 is an integer
#pragma unroll 1
  for (int step = 0; step < 9; step++) {                   // incr = 2^-1 .. 2^-9  (> 1/1024)
    load_row(frac512(early + incr), up);                   // candidates for the next step / the final point
    load_row(frac512(early - incr), dn);
    v2f e, l;
    interp2(early, 2, cur, e, l);
    const float ne = norm2(unpk(e)), nl = norm2(unpk(l));
    const bool goUp = ne < nl, goDn = ne > nl;
    if (active) {
      if (goUp) early += incr;
      else if (goDn) early -= incr;
      else active = false;                                 // "else break" (:695)
      if (active) incr = incr * 0.5f;
    }
    const bool moved = active;                             // row changes only if the index moved
#pragma unroll
    for (int j = 0; j < 12; j++) cur[j] = moved ? (goUp ? up[j] : dn[j]) : cur[j];
  }
  *peakIx = early + 1.0f;                                  // same fractional part as `early`: row = cur
  v2f peak, dummy;
  interp2(*peakIx, 0, cur, peak, dummy);
  return unpk(peak);
}


// ---------------------------------------------------------------------------------------------
// pair_bisect: peakDetect's bisection with TWO lanes per burst -- the even lane (h = 0) evaluates the early
// point of every step, the odd lane (h = 1) the late one: the two 21-tap sums are the only parallelism the
// reference's loop has (each sum must run j = 0..20 in order).  State: early = M-1 + e/512 (all of the
// reference's +-2^-k steps are exact in float).
//   stab  the sinc table in LDS (SincLds below); the row of the NEXT step is fetched after the decision
//         (6 ds_read_b128) instead of gathering both candidates from L2 a step ahead;
//   q[k]  = corr[M - 12 + k + 2h] with the zeros interpolatePoint implies (lag < 0, lag > n-2, :646), held in
//         REGISTERS: floor(early) is fixed after step 0 (early stays inside (M-2, M-1) or (M-1, M)), so one
//         select after step 0 makes every later tap index static.  Tap j of the early point is q[base + j] on the
//         even lane, of the late point q[base + j] on the odd lane (base = floor(early) - (M - 2) = 0 or 1), of
//         the final point q[base + 1 + j] on the even lane.
// Returns interpolatePoint(early + 1) (:699-700) -- the even lane's value counts -- and e.
// ---------------------------------------------------------------------------------------------
struct SincLds { float row[512][24]; };                    // 48 KB

// one workgroup of NT threads copies the table: issue the loads early, store when convenient, then __syncthreads()
template <int NT>
__device__ __forceinline__ void sinc_lds_issue(const TrxTables *__restrict__ T, int tid, float4 (&tv)[3072 / NT]) {
#pragma unroll
  for (int k = 0; k < 3072 / NT; k++) {
    const int ix = tid + NT * k;                           // 3072 float4 = 512 rows x 6
    tv[k] = *reinterpret_cast<const float4 *>(&T->sinc_grid[ix / 6][4 * (ix % 6)]);
  }
}
template <int NT>
__device__ __forceinline__ void sinc_lds_store(SincLds &S, int tid, const float4 (&tv)[3072 / NT]) {
#pragma unroll
  for (int k = 0; k < 3072 / NT; k++) {
    const int ix = tid + NT * k;
    *reinterpret_cast<float4 *>(&S.row[ix / 6][4 * (ix % 6)]) = tv[k];
  }
}

__device__ __forceinline__ cx pair_bisect(const SincLds &S, const cx (&q)[23], int h, int &e_out) {
  auto load_row = [&](int f, float (&s)[24]) {
    const float4 *rw = reinterpret_cast<const float4 *>(S.row[f]);
#pragma unroll
    for (int g = 0; g < 6; g++) {
      const float4 t4 = rw[g];
      s[4 * g] = t4.x; s[4 * g + 1] = t4.y; s[4 * g + 2] = t4.z; s[4 * g + 3] = t4.w;
    }
  };
  int e = 0;
  bool active = true;
  // one early/late decision (:690-697) from this lane's point and its neighbour's
  auto decide = [&](cx pt, int inc) {
    const float mine = norm2(pt);
    const float other = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mine), 0xB1, 0xf, 0xf, true));   // lane ^ 1
    const float ne = h ? other : mine, nl = h ? mine : other;
    if (active) {
      if (ne < nl) e += inc;
      else if (ne > nl) e -= inc;
      else active = false;                                 // "else break" (:695)
    }
  };
  float srow[24];
  // step 0: early = M-1 is an integer (row 0, base 1)
  load_row(0, srow);
  {
    cx pt = mk(0, 0);
#pragma unroll
    for (int j = 0; j < 21; j++) pt = cadd(pt, cmulr(q[1 + j], srow[j]));
    decide(pt, 256);
  }
  // floor(early) = M-2 if the first step went down, M-1 otherwise, and stays there
  cx w[22];
#pragma unroll
  for (int k = 0; k < 22; k++) w[k] = (e < 0) ? q[k] : q[k + 1];
#pragma unroll 1
  for (int inc = 128; inc >= 1; inc >>= 1) {               // increments 2^-2 .. 2^-9
    load_row(e & 511, srow);
    cx pt = mk(0, 0);
#pragma unroll
    for (int j = 0; j < 21; j++) pt = cadd(pt, cmulr(w[j], srow[j]));
    decide(pt, inc);
  }
  load_row(e & 511, srow);                                 // early + 1 has the same fractional part
  cx peak = mk(0, 0);
#pragma unroll
  for (int j = 0; j < 21; j++) peak = cadd(peak, cmulr(w[j + 1], srow[j]));
  e_out = e;
  return peak;
}


// One super-step of the speculative bisection.  State on entry: early = M-1 + e/512, `active`.  Lane hl
// evaluates node hp's early (add 0) or late (add 2) point, or (FIN) one of the 2^NLV candidate final
// points (add 1); the node's index offset, in units of this super-step's smallest increment, and `add`
// are lane constants that come from a table (FusedRel).
template <int NLV, bool FIN>
struct FusedRel {
  static constexpr int NEV = 2 * ((1 << NLV) - 1);
  static constexpr int NTOT = NEV + (FIN ? (1 << NLV) : 0);
  int v[64];                                               // (offset << 2) | add; lanes past NTOT: the root's early point
  constexpr FusedRel() : v() {
    for (int hl = 0; hl < 64; hl++) {
      int hp = 1, add = 0;
      if (hl < NEV) { hp = (hl >> 1) + 1; add = 2 * (hl & 1); }
      else if (FIN && hl < NTOT) { hp = (1 << NLV) + (hl - NEV); add = 1; }
      int l = 0;
      while ((hp >> (l + 1)) != 0) l++;                    // depth of the node below the super-step's root
      int r = 0;
      for (int k = 0; k < l; k++) r += (((hp >> (l - 1 - k)) & 1) ? 1 : -1) * (1 << (NLV - 1 - k));
      v[hl] = r * 4 + add;
    }
  }
};
__device__ __constant__ const FusedRel<5, false> kFusedRel5;
__device__ __constant__ const FusedRel<4, true> kFusedRel4F;
__device__ __constant__ const FusedRel<4, false> kFusedRel4;
__device__ __constant__ const FusedRel<1, true> kFusedRel1F;
__device__ __constant__ const FusedRel<3, false> kFusedRel3;
__device__ __constant__ const FusedRel<2, false> kFusedRel2;
__device__ __constant__ const FusedRel<1, false> kFusedRel1;

__device__ __forceinline__ void fused_row(const TrxTables *__restrict__ T, int e_lane, float (&s)[24]) {
  const float4 *row = reinterpret_cast<const float4 *>(T->sinc_grid[e_lane & 511]);   // frac(ix)*512
#pragma unroll
  for (int q = 0; q < 6; q++) {
    const float4 v = row[q];
    s[4 * q] = v.x; s[4 * q + 1] = v.y; s[4 * q + 2] = v.z; s[4 * q + 3] = v.w;
  }
}

// interpolatePoint (:650-657) at early + add: tap 0 sits at loc[floor(ix) - 10 - (M - 12)]
__device__ __forceinline__ cx fused_point(const cx *loc, int e_lane, int add, const float (&s)[24]) {
  const int base = 1 + (e_lane >> 9) + add;
  cx pt = mk(0, 0);
#pragma unroll
  for (int j0 = 0; j0 < 21; j0 += 7) {                     // (chunked: keeps the LDS reads from piling up in VGPRs)
    cx lv[7];
#pragma unroll
    for (int j = 0; j < 7; j++) lv[j] = loc[base + j0 + j];
#pragma unroll
    for (int j = 0; j < 7; j++) pt = cadd(pt, cmulr(lv[j], s[j0 + j]));   // j ascending
    __builtin_amdgcn_sched_barrier(0);
  }
  return pt;
}

// Replays the NLV early/late decisions (:690-697) along the path the reference takes, from the lanes'
// powers: even lane 2(h-1) holds node h's early point, its odd neighbour the late one.  inc0: first
// increment (1/512 units).  FIN: `peak` = the candidate final point of the leaf reached.
template <int LPB, int NLV, bool FIN>
__device__ __forceinline__ void fused_decide(cx pt, int lane, int inc0, int &e, bool &active, cx &peak) {
  const float pw = norm2(pt);
  const float other = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(pw), 0xB1, 0xf, 0xf, true));   // lane ^ 1
  const unsigned long long mu = __builtin_amdgcn_ballot_w64(pw < other);   // even lanes: early < late
  const unsigned long long md = __builtin_amdgcn_ballot_w64(pw > other);
  typedef typename std::conditional<LPB == 64, unsigned long long, unsigned>::type mask_t;
  const int sh = lane & (64 - LPB);                        // first lane of this burst
  const mask_t bu = (mask_t)(mu >> sh), bd = (mask_t)(md >> sh);
  int hq = 1;
#pragma unroll
  for (int k = 0; k < NLV; k++) {
    const int ix = 2 * (hq - 1);
    const bool up = (bu >> ix) & 1, dn = (bd >> ix) & 1;
    const int step = inc0 >> k;
    if (active) {
      if (up) { e += step; hq = 2 * hq + 1; }
      else if (dn) { e -= step; hq = 2 * hq; }
      else active = false;                                 // "else break" (:695): e stays put from here on
    }
  }
  if (FIN) {
    const int src = (lane & ~(LPB - 1)) + 2 * ((1 << NLV) - 1) + ((hq - (1 << NLV)) & ((1 << NLV) - 1));
    peak = mk(__shfl(pt.r, src, 64), __shfl(pt.i, src, 64));
  }
}

// analyzeTrafficBurst after peakDetect (:959-1000, k_tsc_peak's arithmetic) plus energyDetect's decision,
// for one burst per LPB lanes: every lane of the burst gets the same results.  State from the
// bisection: peak index = M + e/512.  pw_at(lag) = |corr[lag]|^2, or 0 outside [0, n).  V: NV floats of
// the burst's scratch (16-byte aligned).
template <int SPS, int LPB, typename PWF>
__device__ __forceinline__ void fused_tail(PWF pw_at, float *V, int hl, int M, int e, cx peak, bool good, float energy,
                                           cx gain_inv, float mid_toa, float detect_thresh, float energy_thresh,
                                           cx &amp, float &toa, bool &detected, bool &energy_ok) {
  constexpr int NL = 36 * SPS, NE = 20 * SPS, NV = 2 * (3 * SPS + 1);
  const float early = (float)(M - 1) + (float)e * 0.001953125f;   // exact: the reference's +-2^-k steps are exact too
  toa = early + 1.0f;                                      // :699
  amp = peak;
  detected = false;
  energy_ok = good && (energy_thresh < 0.0f || energy / (float)(unsigned)NE > energy_thresh * energy_thresh);
  const bool sane = !(toa < 0.0f) && !(toa > (float)NL) && good;
  const int pk = sane ? (int)rintf(toa) : 0;
  // valley terms in the reference's order (:971-980): i = 2sps..5sps, (peak - i) then (peak + i);
  // terms the reference skips (index < 0 or >= n) come back as +0: adding +0 to a sum of
  // non-negative terms changes nothing, and numRms is counted arithmetically below.
#pragma unroll
  for (int t0 = 0; t0 < NV; t0 += LPB) {
    const int tt = t0 + hl;
    if (tt < NV) {
      const int i = 2 * SPS + (tt >> 1);
      V[tt] = pw_at((tt & 1) ? pk + i : pk - i);
    }
  }
  wave_lds_fence();
  if (sane) {
    float valley = 0.0f;
    const float4 *V4 = reinterpret_cast<const float4 *>(V);
#pragma unroll
    for (int q = 0; q < (NV + 3) / 4; q++) {
      const float4 t = V4[q];
      valley = valley + t.x;
      if (4 * q + 1 < NV) valley = valley + t.y;
      if (4 * q + 2 < NV) valley = valley + t.z;
      if (4 * q + 3 < NV) valley = valley + t.w;
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
}


}  // namespace
