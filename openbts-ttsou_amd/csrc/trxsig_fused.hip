// ---------------------------------------------------------------------------------------------
// trxsig_fused.hip -- single-launch alternates of the normal-burst path.
//
// k_normal_fused: the whole normal-burst leg of pullRadioVector (energyDetect, analyzeTrafficBurst,
//   demodulateBurst) for one burst per LPB lanes (LPB = 64: one wave per burst, LPB = 32: two bursts
//   per wave), reading the burst from HBM exactly once and writing only the results.
//
//   The serial part of the reference -- peakDetect's 9-step early/late bisection (:684-701) -- is
//   turned into 2 (LPB = 64) or 3 (LPB = 32) parallel "super-steps": the bisection is a binary
//   decision tree whose node at depth d is reached with a known index offset, so the lanes evaluate
//   interpolatePoint at the early and late points of EVERY node of the next NLV levels at once
//   (2*(2^NLV - 1) points), and the decisions are then replayed along the one path the reference
//   takes.  The last super-step also evaluates the 2^NLV possible final points.  Each point is the
//   reference's own 21-term sum in the reference's order, so the chosen path and every value on it
//   are bit-identical; the points off the path are discarded.
//
//   Per-burst scratch (window, correlation, norms, ...) lives in the burst's LDS slot and is
//   overlaid by the demodulator's staging area once detection is done.  No workgroup barrier.
// ---------------------------------------------------------------------------------------------
#ifndef TRX_FUSED_WAVES
#define TRX_FUSED_WAVES 4
#endif

// Numerical contract: see trxsig_dev.h / DESIGN.md (every float32 operation is the reference's, in the
// reference's order; built with -ffp-contract=off).
#include "trxsig_bisect.h"
#include "trxsig_corr.h"
#include "trxsig_demod.h"

namespace {

template <int SPS, int LPB>
__global__ __launch_bounds__(64 * TRX_FUSED_WAVES) void k_normal_fused(
    const TrxTables *__restrict__ T, const cx *__restrict__ samples, const int32_t *__restrict__ offset,
    const int32_t *__restrict__ length, int B, TapArg taps, cx gain_inv, float mid_toa, float detect_thresh,
    float energy_thresh, uint8_t *__restrict__ flags, cx *__restrict__ amp_out, float *__restrict__ toa_out,
    float *__restrict__ avgpwr_out, float *__restrict__ soft, uint8_t *__restrict__ hard, int nsoft, int stride) {
  typedef FusedGeom<SPS, LPB> G;
  __shared__ __attribute__((aligned(16))) cx region[TRX_FUSED_WAVES * G::BPW][G::REG];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hb = lane / LPB, hl = lane % LPB;
  int b = (blockIdx.x * TRX_FUSED_WAVES + wave) * G::BPW + hb;
  if (LPB == 64) b = __builtin_amdgcn_readfirstlane(b);
  cx *R = region[wave * G::BPW + hb];
  cx *W = R + G::O_W, *Cc = R + G::O_C, *loc = R + G::O_LOC;
  float *E = reinterpret_cast<float *>(R + G::O_E);
  float *V = reinterpret_cast<float *>(R + G::O_V);

  const bool live = b < B;
  int off = 0, N = 0;
  if (live) { off = offset[b]; N = length[b]; }
  const bool good = live && (off >= 0) && (N >= 92 * SPS) && (N <= 157 * SPS) && (N % SPS == 0);
  const cx *xb = samples + (good ? off : 0);

  // ---- the burst's only trip through HBM: pair q = hl + LPB*i holds samples 2q, 2q+1 ----
  float4 v[G::NLD];
  {
    const bool wide = (off & 1) == 0;
    const float4 *xv = reinterpret_cast<const float4 *>(xb);
#pragma unroll
    for (int i = 0; i < G::NLD; i++) {
      const int q = hl + LPB * i, n0 = 2 * q;
      float4 t = make_float4(0, 0, 0, 0);
      if (good && n0 + 1 < N) {
        if (wide) t = xv[q];
        else { const cx a = xb[n0], c = xb[n0 + 1]; t = make_float4(a.r, a.i, c.r, c.i); }
      } else if (good && n0 < N) {
        const cx a = xb[n0]; t = make_float4(a.r, a.i, 0, 0);
      }
      v[i] = t;
    }
  }
  cx tap[16];
#pragma unroll
  for (int k = 0; k < 16; k++) tap[k] = mk(taps.v[2 * k], taps.v[2 * k + 1]);

  // ---- zero-padded correlation window w = burst[56*sps, 92*sps) and |x|^2 of the energy window ----
  for (int q = hl; q < G::FRONT; q += LPB) W[q] = mk(0, 0);
  for (int q = G::FRONT + G::NL + hl; q < G::WLEN; q += LPB) W[q] = mk(0, 0);
  for (int q = hl; q < G::PADC; q += LPB) { Cc[q] = mk(0, 0); Cc[G::PADC + G::NL + q] = mk(0, 0); }
#pragma unroll
  for (int i = 0; i < G::NLD; i++) {
    constexpr int W0 = 56 * SPS, W1 = 92 * SPS;            // both even: a pair is in or out as a whole
    const int n0 = 2 * (hl + LPB * i);
    if (2 * LPB * i < W1 && 2 * LPB * (i + 1) > W0) {
      if (n0 >= W0 && n0 < W1) *reinterpret_cast<float4 *>(W + G::FRONT + n0 - W0) = v[i];
    }
    if (2 * LPB * i < G::NE) {
      if (n0 < G::NE)
        *reinterpret_cast<float2 *>(E + n0) = make_float2(norm2(mk(v[i].x, v[i].y)), norm2(mk(v[i].z, v[i].w)));
    }
  }
  wave_lds_fence();

  // ---- energyDetect: energy += norm2(x[i]) strictly in order (:925-928); broadcast reads ----
  float energy = 0.0f;
  {
    // (in chunks, each pinned: otherwise hipcc keeps all NE norms -- 80 VGPRs -- live across the correlation)
    const float4 *E4 = reinterpret_cast<const float4 *>(E);
#pragma unroll
    for (int c4 = 0; c4 < G::NE / 4; c4 += 4) {
      float4 ev[4];
#pragma unroll
      for (int q = 0; q < 4; q++) ev[q] = (c4 + q < G::NE / 4) ? E4[c4 + q] : make_float4(0, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (c4 + q < G::NE / 4) {
          energy = energy + ev[q].x; energy = energy + ev[q].y; energy = energy + ev[q].z; energy = energy + ev[q].w;
        }
      }
      asm volatile("" : "+v"(energy));                     // pin the chain here (else it is sunk to its use, norms and all)
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- correlation with the 16 non-zero taps, k descending = j ascending (k_tsc_corr's arithmetic) ----
  float bestP = 0.0f;
  int bestT = -1;
  {
    const int g = hl / SPS, p = hl % SPS;
    const bool owner = g < G::GA;
    const int base = SPS * G::CG * (owner ? g : 0) + p;
    cx sv[G::NSV];
#pragma unroll
    for (int u = 0; u < G::NSV; u++) sv[u] = W[base + SPS * u];
#pragma unroll
    for (int i = 0; i < G::CG; i++) {
      cx acc = mk(0, 0);
#pragma unroll
      for (int k = 15; k >= 0; k--) acc = cadd(acc, cmul(sv[i + k], tap[k]));
      const int t = base + SPS * i;
      if (owner && t < G::NL) {
        Cc[G::PADC + t] = acc;
        const float pw = norm2(acc);
        if (pw > bestP) { bestP = pw; bestT = t; }         // strict >, first maximum (:675)
      }
    }
  }
  // the first super-step's points do not depend on the data (early starts at M-1): fetch its sinc rows
  // now, under the argmax reduction
  const int relA = (LPB == 64) ? kFusedRel5.v[hl] : kFusedRel4.v[hl];
  const int eA = (relA >> 2) * ((LPB == 64) ? 16 : 32);
  float rowA[24];
  fused_row(T, eA, rowA);
#pragma unroll
  for (int m = 1; m < LPB; m <<= 1) {                      // larger power wins, equal power -> smaller lag
    const float oP = __shfl_xor(bestP, m, 64);
    const int oT = __shfl_xor(bestT, m, 64);
    const bool take = (oP > bestP) || (oP == bestP && oT >= 0 && (bestT < 0 || oT < bestT));
    if (take) { bestP = oP; bestT = oT; }
  }
  const int M = bestT;
  wave_lds_fence();

  // ---- lags M-12 .. M+11 as interpolatePoint sees them (never the last sample, :646) ----
  if (hl < 26) {
    const int lag = M - 12 + hl;
    loc[hl] = (hl >= 24 || lag > G::NL - 2) ? mk(0, 0) : Cc[G::PADC + lag];
  }
  wave_lds_fence();

  // ---- peakDetect's bisection, speculated (see the header) ----
  int e = 0;                                               // early = M-1 + e/512
  bool active = true;
  cx peak = mk(0, 0);
  if constexpr (LPB == 64) {
    const cx ptA = fused_point(loc, eA, relA & 3, rowA);                    // levels 1-5: +-256 .. +-16
    fused_decide<LPB, 5, false>(ptA, lane, 256, e, active, peak);
    const int relB = kFusedRel4F.v[hl], eB = e + (relB >> 2);               // levels 6-9: +-8 .. +-1, and the finals
    float rowB[24];
    fused_row(T, eB, rowB);
    const cx ptB = fused_point(loc, eB, relB & 3, rowB);
    fused_decide<LPB, 4, true>(ptB, lane, 8, e, active, peak);
  } else {
    const cx ptA = fused_point(loc, eA, relA & 3, rowA);                    // levels 1-4: +-256 .. +-32
    fused_decide<LPB, 4, false>(ptA, lane, 256, e, active, peak);
    const int relB = kFusedRel4.v[hl], eB = e + 2 * (relB >> 2);            // levels 5-8: +-16 .. +-2
    float rowB[24];
    fused_row(T, eB, rowB);
    const cx ptB = fused_point(loc, eB, relB & 3, rowB);
    fused_decide<LPB, 4, false>(ptB, lane, 16, e, active, peak);
    const int relC = kFusedRel1F.v[hl], eC = e + (relC >> 2);               // level 9: +-1, and the finals
    float rowC[24];
    fused_row(T, eC, rowC);
    const cx ptC = fused_point(loc, eC, relC & 3, rowC);
    fused_decide<LPB, 1, true>(ptC, lane, 1, e, active, peak);
  }
  if (!active) {
    // the reference left its loop on equal powers (:695): the peak is interpolatePoint(early + 1)
    // at the index where it stopped, which no lane has speculated.  Rare (e.g. an all-zero window).
    const float4 *row = reinterpret_cast<const float4 *>(T->sinc_grid[e & 511]);
    float s[24];
#pragma unroll
    for (int q = 0; q < 6; q++) {
      const float4 r4 = row[q];
      s[4 * q] = r4.x; s[4 * q + 1] = r4.y; s[4 * q + 2] = r4.z; s[4 * q + 3] = r4.w;
    }
    const int base = 2 + (e >> 9);
    cx pt = mk(0, 0);
#pragma unroll
    for (int j = 0; j < 21; j++) pt = cadd(pt, cmulr(loc[base + j], s[j]));
    peak = pt;
  }
  cx amp;
  float toa;
  bool detected, energy_ok;
  fused_tail<SPS, LPB>([&](int lag) { return norm2(Cc[G::PADC + lag]); }, V, hl, M, e, peak, good, energy, gain_inv, mid_toa,
                       detect_thresh, energy_thresh, amp, toa, detected, energy_ok);

  if (live && hl == 0) {
    uint8_t fl = 0;
    if (!good) fl = TRXSIG_F_BADLEN;
    else fl = (energy_ok ? TRXSIG_F_ENERGY : 0) | (detected ? TRXSIG_F_DETECT : 0);
    flags[b] = fl;
    amp_out[b] = amp;
    toa_out[b] = toa;
    if (avgpwr_out) avgpwr_out[b] = good ? energy / (float)(unsigned)G::NE : 0.0f;
  }
  if (nsoft <= 0 || !live) return;                         // (LPB = 32: a dead upper half has nothing to write)

  // ---- demodulateBurst (k_demod's arithmetic) from the samples still in registers ----
  float *sb = soft + (size_t)b * stride;
  uint8_t *hbp = hard ? hard + (size_t)b * stride : nullptr;
  const bool lane_owner = G::OPL * hl < 148;
  const int m0 = G::OPL * (lane_owner ? hl : 0);
  if (!detected) {
#pragma unroll
    for (int i = 0; i < G::OPL; i++) {
      const int m = m0 + i;
      if (lane_owner && m < nsoft) { sb[m] = 0.0f; if (hbp) hbp[m] = 0; }
    }
    return;
  }
  fused_demod<SPS, LPB>(T, R, v, N, amp, toa, hl, sb, hbp, nsoft, [] {}, nullptr, nullptr);
}



// ---------------------------------------------------------------------------------------------
// k_normal_quad: the normal-burst leg in one kernel with FOUR bursts per wave.
//   Phase 1 (16 lanes per burst, k_tsc_corr's code): window + energy loads, correlation, argmax; the
//     correlation stays in the burst's LDS row.
//   Phase 2 (16 lanes per burst): peakDetect's bisection speculated three levels at a time (14 of the
//     16 lanes evaluate the early/late points of the next 7 tree nodes, fused_point/fused_decide),
//     3 super-steps + the final point; then analyzeTrafficBurst's tail.  Per-burst results live in
//     the registers of the burst's lanes.
//   Phase 3 (the whole wave per burst, one burst after the other): demodulateBurst (fused_demod) with
//     the next burst's samples already in flight.  Its staging area overlays the four dead rows.
//   The uniform per-burst work is shared by four bursts and the correlation runs with every lane
//   busy, which is what the wave-per-burst kernel above cannot do; the price is that the window and
//   the energy samples are read twice (the second time from L2).  No workgroup barrier.
// ---------------------------------------------------------------------------------------------
template <int SPS, unsigned TAPCLS, bool DEMOD, bool TOL = false>
__global__ __launch_bounds__(256) void k_normal_quad(
    const TrxTables *__restrict__ T, const cx *__restrict__ samples, const int32_t *__restrict__ offset,
    const int32_t *__restrict__ length, int B, TapArg taps, cx gain_inv, float mid_toa, float detect_thresh,
    float energy_thresh, uint8_t *__restrict__ flags, cx *__restrict__ amp_out, float *__restrict__ toa_out,
    float *__restrict__ avgpwr_out, float *__restrict__ soft, uint8_t *__restrict__ hard, int nsoft, int stride) {
  typedef CorrGeom<SPS> G;
  typedef FusedGeom<SPS, 64> F;
  typedef typename F::D D;
  static_assert(G::WPAD - G::NL >= 26 + F::NV / 2 + 1, "row has no room for the bisection scratch");
  static_assert(4 * G::WPAD >= D::U, "four rows must hold the demodulator's staging area");
  static_assert(8 * G::WPAD >= 4 * G::NE, "the energy norms are staged in the row itself");
  __shared__ __attribute__((aligned(16))) cx rows[16][G::WPAD];
  // sinc rows f = 0, 16, .., 496: all that the first two super-steps of the bisection can ask for
  // (their nodes sit on multiples of 16/512), so only the last super-step and the final point gather from L2
  __shared__ __attribute__((aligned(16))) float stab[32][24];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row = lane >> 4, r = lane & 15;
  const int slot = wave * 4 + row;
  cx *W = rows[slot];
  {
    float tv[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { const int ix = threadIdx.x * 3 + k; tv[k] = T->sinc_grid[16 * (ix / 24)][ix % 24]; }
#pragma unroll
    for (int k = 0; k < 3; k++) { const int ix = threadIdx.x * 3 + k; stab[ix / 24][ix % 24] = tv[k]; }
  }

  // ---- phase 1 ----
  int M;
  float energy;
  CorrIn<SPS> in;
  {
    cx tap[16];
#pragma unroll
    for (int k = 0; k < 16; k++) tap[k] = mk(taps.v[2 * k], taps.v[2 * k + 1]);
    corr_issue<SPS>(in, blockIdx.x * 16 + slot, B, r, samples, offset, length);
    corr_round<SPS, false, true, TAPCLS>(in, W, reinterpret_cast<float4 *>(W), lane, r, tap, nullptr, 0, M, energy);
  }
  const bool live = in.live, good = in.good;
  const int b = in.b;
  __syncthreads();                                         // the sinc rows of all four waves are in stab (the only barrier)

  // ---- phase 2 ----
  cx *loc = W + G::NL;                                     // lags M-12 .. M+11 as interpolatePoint sees them (:646)
  float *V = reinterpret_cast<float *>(W + G::NL + 26);
#pragma unroll
  for (int j0 = 0; j0 < 26; j0 += 16) {
    const int j = j0 + r;
    if (j < 26) {
      const int lag = M - 12 + j;
      loc[j] = (j >= 24 || lag < 0 || lag > G::NL - 2) ? mk(0, 0) : W[lag];
    }
  }
  wave_lds_fence();
  int e = 0;                                               // early = M-1 + e/512
  asm volatile("" : "+v"(e));                              // (opaque: keeps the first sinc-row fetch from being hoisted above phase 1)
  bool active = true;
  cx peak = mk(0, 0);
  {
    const int rel = kFusedRel3.v[r];
#pragma unroll
    for (int st = 0; st < 3; st++) {                       // increments 256,128,64 | 32,16,8 | 4,2,1
      const int inc_last = 64 >> (3 * st);
      const int el = e + (rel >> 2) * inc_last;
      float srow[24];
      if (st < 2) {                                        // nodes on multiples of 16/512: the LDS copy
        const float4 *rw = reinterpret_cast<const float4 *>(stab[(el & 511) >> 4]);
#pragma unroll
        for (int q = 0; q < 6; q++) {
          const float4 t4 = rw[q];
          srow[4 * q] = t4.x; srow[4 * q + 1] = t4.y; srow[4 * q + 2] = t4.z; srow[4 * q + 3] = t4.w;
        }
      } else {
        fused_row(T, el, srow);
      }
      const cx pt = fused_point(loc, el, rel & 3, srow);
      fused_decide<16, 3, false>(pt, lane, 4 * inc_last, e, active, peak);
    }
    // the loop ended (all nine steps, or the reference's `break` on equal powers): the peak is
    // interpolatePoint(early + 1) at the index where it stopped (:699-700)
    float srow[24];
    fused_row(T, e, srow);
    peak = fused_point(loc, e, 1, srow);
    asm volatile("" : "+v"(peak.r), "+v"(peak.i));         // (finished here: not to be interleaved with the tail)
  }
  cx amp;
  float toa;
  bool detected, energy_ok;
  fused_tail<SPS, 16>([&](int lag) { return (lag < 0 || lag >= G::NL) ? 0.0f : norm2(W[lag]); }, V, r, M, e, peak, good,
                      energy, gain_inv, mid_toa, detect_thresh, energy_thresh, amp, toa, detected, energy_ok);
  if (live && r == 0) {
    uint8_t fl = 0;
    if (!good) fl = TRXSIG_F_BADLEN;
    else fl = (energy_ok ? TRXSIG_F_ENERGY : 0) | (detected ? TRXSIG_F_DETECT : 0);
    flags[b] = fl;
    amp_out[b] = amp;
    toa_out[b] = toa;
    if (avgpwr_out) avgpwr_out[b] = good ? energy / (float)(unsigned)G::NE : 0.0f;
  }
  if (!DEMOD || nsoft <= 0) return;                        // DEMOD = false: detection only (k_demod follows)

  // ---- phase 3 ----
  // delayVector's taps for each burst's TOA (fused_demod's arithmetic), fetched now by the burst's own
  // lanes -- lane r holds taps r and r+16 -- so that phase 3 finds them in registers
  float tap_lo, tap_hi;
  {
    const float delay = -toa;
    const float frac = delay - (float)(int)floorf(delay);
    const float f512 = frac * 512.0f;
    const int f = (int)f512;
    if ((float)f == f512) {                                // on the 1/512 grid (always, after peakDetect)
      tap_lo = T->sinc_grid[f & 511][r];
      tap_hi = T->sinc_grid[f & 511][16 + (r & 7)];
    } else {
      tap_lo = dev_sinc(T->sinT, TRX_PI_F * ((float)(r - 10) - frac));            // :588
      tap_hi = dev_sinc(T->sinT, TRX_PI_F * ((float)(16 + (r & 7) - 10) - frac));
    }
  }
  cx rvl[F::OPL];                                          // the lane's reverse-rotation values (same for every burst)
#pragma unroll
  for (int i = 0; i < F::OPL; i++) rvl[i] = T->rev[SPS * (F::OPL * (F::OPL * lane < 148 ? lane : 0) + i)];
  wave_lds_fence();                                        // the rows are dead from here on
  cx *P = rows[wave * 4];
  const int b0 = blockIdx.x * 16 + wave * 4;
  auto fetch = [&](int rr, int lane, float4 (&v)[F::NLD], int &N, bool &det) {
    // per-burst scalars come from lane 16*rr; the samples: pair q = lane + 64*i holds samples 2q, 2q+1
    det = __builtin_amdgcn_readlane((int)(detected && live), 16 * rr) != 0;
    N = __builtin_amdgcn_readlane(in.good ? length[in.live ? b : 0] : 0, 16 * rr);
    const int off = __builtin_amdgcn_readlane(in.good ? offset[in.live ? b : 0] : 0, 16 * rr);
    const cx *xb = samples + off;
    const bool wide = (off & 1) == 0;
    const float4 *xv = reinterpret_cast<const float4 *>(xb);
#pragma unroll
    for (int i = 0; i < F::NLD; i++) {
      const int q = lane + 64 * i, n0 = 2 * q;
      float4 t = make_float4(0, 0, 0, 0);
      if (det && n0 + 1 < N) {
        if (wide) t = xv[q];
        else { const cx a = xb[n0], c = xb[n0 + 1]; t = make_float4(a.r, a.i, c.r, c.i); }
      } else if (det && n0 < N) {
        const cx a = xb[n0]; t = make_float4(a.r, a.i, 0, 0);
      }
      v[i] = t;
    }
  };
  float4 v[F::NLD];
  int N;
  bool det;
  fetch(0, lane, v, N, det);
#pragma unroll 1
  for (int rr = 0; rr < 4; rr++) {
    // (opaque copy of the lane id: otherwise every per-lane address of the loop body is hoisted out of
    //  the loop and parked in VGPRs across phases -- 30 registers and spills)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    // the next burst's samples are requested before this one is touched (second register set)
    float4 vn[F::NLD];
    int Nn = 0;
    bool detn = false;
    if (rr < 3) fetch(rr + 1, ln, vn, Nn, detn);
    const int bb = b0 + rr;
    if (bb < B) {
      float *sb = soft + (size_t)bb * stride;
      uint8_t *hbp = hard ? hard + (size_t)bb * stride : nullptr;
      if (det) {
        const cx a = mk(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(amp.r), 16 * rr)),
                        __int_as_float(__builtin_amdgcn_readlane(__float_as_int(amp.i), 16 * rr)));
        const float ta = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(toa), 16 * rr));
        float tp[21];
#pragma unroll
        for (int j = 0; j < 21; j++)
          tp[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(j < 16 ? tap_lo : tap_hi), 16 * rr + (j & 15)));
        // (TOL: trxsig_set_soft_mode(TRXSIG_SOFT_TOLERANCE) -- the rearranged form unless this burst has to be exact)
        if (!(TOL && fused_demod_tol<SPS>(T, P, v, N, a, ta, ln, sb, hbp, nsoft)))
          fused_demod<SPS, 64>(T, P, v, N, a, ta, ln, sb, hbp, nsoft, [] {}, tp, rvl);
        wave_lds_fence();                                  // staging reads done before the next burst overwrites it
      } else {
        for (int m = ln; m < nsoft; m += 64) { sb[m] = 0.0f; if (hbp) hbp[m] = 0; }
      }
    }
#pragma unroll
    for (int i = 0; i < F::NLD; i++) v[i] = vn[i];
    N = Nn;
    det = detn;
  }
}



}  // namespace

template <int S, int LPB>
static void launch_normal_fused(hipStream_t st, const TrxTables *dT, const TrxTables *hT, const trx_c32 *samples,
                                const int32_t *off, const int32_t *len, int B, int tsc, float detect_thresh,
                                float energy_thresh, uint8_t *flags, trx_c32 *amp, float *toa, float *avgpwr,
                                float *soft, uint8_t *hard, int nsoft, int stride, int generic_taps, int tol) {
  TapArg ta;
  for (int k = 0; k < 16; k++) { ta.v[2 * k] = hT->mid_ctap[tsc][k].r; ta.v[2 * k + 1] = hT->mid_ctap[tsc][k].i; }
  // gain.inv() (Complex.h:154-160) in the reference's float arithmetic; this file is built with -ffp-contract=off
  const trx_c32 g = hT->mid_gain[tsc];
  const float n = g.i * g.i + g.r * g.r;
  trx_c32 ginv; ginv.r = g.r / n; ginv.i = -g.i / n;
  if (LPB == 16) {
    const dim3 qgrid((B + 15) / 16), qblock(256);
#define TRX_QUAD_ARGS dT, samples, off, len, B, ta, ginv, hT->mid_toa[tsc], detect_thresh, energy_thresh, flags, amp, toa, \
                      avgpwr, soft, hard, nsoft, stride
    const bool spec = !generic_taps && tap_classes(hT, tsc) == TapPattern<S>::value;
    if (nsoft > 0 && tol) {
      if (spec) k_normal_quad<S, TapPattern<S>::value, true, true><<<qgrid, qblock, 0, st>>>(TRX_QUAD_ARGS);
      else k_normal_quad<S, TRX_TAPS_GENERIC, true, true><<<qgrid, qblock, 0, st>>>(TRX_QUAD_ARGS);
    } else if (nsoft > 0) {
      if (spec) k_normal_quad<S, TapPattern<S>::value, true><<<qgrid, qblock, 0, st>>>(TRX_QUAD_ARGS);
      else k_normal_quad<S, TRX_TAPS_GENERIC, true><<<qgrid, qblock, 0, st>>>(TRX_QUAD_ARGS);
    } else {
      if (spec) k_normal_quad<S, TapPattern<S>::value, false><<<qgrid, qblock, 0, st>>>(TRX_QUAD_ARGS);
      else k_normal_quad<S, TRX_TAPS_GENERIC, false><<<qgrid, qblock, 0, st>>>(TRX_QUAD_ARGS);
    }
#undef TRX_QUAD_ARGS
  } else {
    constexpr int L = LPB == 16 ? 64 : LPB;
    constexpr int per_wg = TRX_FUSED_WAVES * (64 / L);
    k_normal_fused<S, L><<<dim3((B + per_wg - 1) / per_wg), dim3(64 * TRX_FUSED_WAVES), 0, st>>>(
        dT, samples, off, len, B, ta, ginv, hT->mid_toa[tsc], detect_thresh, energy_thresh, flags, amp, toa, avgpwr, soft,
        hard, nsoft, stride);
  }
}

hipError_t trx_launch_normal_fused(hipStream_t st, int sps, int lanes_per_burst, const TrxTables *dT, const TrxTables *hT,
                                   const trx_c32 *samples, const int32_t *off, const int32_t *len, int B, int tsc,
                                   float detect_thresh, float energy_thresh, uint8_t *flags, trx_c32 *amp, float *toa,
                                   float *avgpwr, float *soft, uint8_t *hard, int nsoft, int stride,
                                   int generic_taps, TrxProfiler *prof, int soft_tolerance) {
  if (B <= 0) return hipSuccess;
  if (nsoft > 148 || (lanes_per_burst != 64 && lanes_per_burst != 32 && lanes_per_burst != 16)) return hipErrorInvalidValue;
  if (prof) prof->begin(TRXSIG_K_NORMAL_FUSED, st);
#define TRX_FUSED_CASE(S)                                                                                              \
  case S:                                                                                                              \
    if (lanes_per_burst == 64)                                                                                         \
      launch_normal_fused<S, 64>(st, dT, hT, samples, off, len, B, tsc, detect_thresh, energy_thresh, flags, amp, toa, \
                                 avgpwr, soft, hard, nsoft, stride, generic_taps, soft_tolerance);                                   \
    else if (lanes_per_burst == 16)                                                                                    \
      launch_normal_fused<S, 16>(st, dT, hT, samples, off, len, B, tsc, detect_thresh, energy_thresh, flags, amp, toa, \
                                 avgpwr, soft, hard, nsoft, stride, generic_taps, soft_tolerance);                                   \
    else                                                                                                               \
      launch_normal_fused<S, 32>(st, dT, hT, samples, off, len, B, tsc, detect_thresh, energy_thresh, flags, amp, toa, \
                                 avgpwr, soft, hard, nsoft, stride, generic_taps, soft_tolerance);                                   \
    break;
  switch (sps) {
    TRX_FUSED_CASE(1)
    TRX_FUSED_CASE(2)
    TRX_FUSED_CASE(4)
    default: return hipErrorInvalidValue;
  }
#undef TRX_FUSED_CASE
  if (prof) prof->end(TRXSIG_K_NORMAL_FUSED, st);
  return hipGetLastError();
}

