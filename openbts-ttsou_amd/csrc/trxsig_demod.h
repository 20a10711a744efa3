// trxsig_demod.h -- demodulateBurst on the device: geometry of the LDS staging area, the two FIR forms
// (demod_core: a lane owns soft bits m, m+64, m+128; fused_demod: a lane owns consecutive soft bits whose
// windows share staged words) and the k_demod kernel template (instantiated by trxsig_normal.hip for the
// burst path and by trxsig_eq.hip for the equaliser's delayVector step).
// Numerical contract: see trxsig_dev.h / DESIGN.md (every float32 operation is the reference's, in the
// reference's order; built with -ffp-contract=off).
#pragma once
#include "trxsig_dev.h"

namespace {

// ---------------------------------------------------------------------------------------------
// k_demod: demodulateBurst (sigProcLib.cpp:1056-1097): scaleVector(1/amp) -> delayVector(-TOA) ->
//   GMSKReverseRotate -> decimateVector(sps) -> vectorSlicer.  One wave per burst, no workgroup
//   barrier (each wave owns its LDS slice).
//
// Only the decimated outputs are ever looked at, so the 21-tap fractional-delay FIR (:584-590) is
// evaluated at t = sps*m - intOffset only.  The scaled burst is staged in LDS *already shifted by
// the integer delay* (sample n at position u = n + intOffset + C) and in polyphase order (position
// u at [u % sps][u / sps]): output m then reads positions sps*m + 10 - j + C, whose phase and
// offset are compile-time constants, so every tap is one ds_read_b64 at base+immediate and the 64
// lanes of a read are contiguous (no bank conflicts).  The 21 real taps are wave-uniform and live
// in SGPRs: from the sinc grid when -TOA lies on the 1/512 grid (always, after peakDetect), else
// computed with the reference's table sinc.
// ---------------------------------------------------------------------------------------------
template <int SPS, int NSMAX>
struct DemodGeom {
  // Output m reads positions SPS*m + (10 - j) + C, j = 0..20, so with C >= 10 every read lands in
  // [0, SPS*(NSMAX-1) + 20 + C] whatever the delay is; samples shifted outside that range are never
  // read and are simply not written.  NSMAX = 148 (the soft bits that go on the wire) keeps the
  // staged burst under 5 KB, i.e. 32 waves (bursts in flight) per CU instead of 28.
  static constexpr int C = 12;                                   // position of sample 0 at intOffset 0 (multiple of 4)
  static constexpr int QLEN = NSMAX + (20 + C) / SPS + 1;        // entries per phase
  static constexpr int U = SPS * QLEN;                           // positions
};

// everything of demodulateBurst after the burst's loads have been issued (v[] = the 16-byte loads of
// the `wide` path, in flight): scale, stage, filter at the decimated instants, rotate, slice, store
template <int SPS, bool RAW, int NSMAX, typename SMP = SmpC32>
__device__ __forceinline__ void demod_core(const TrxTables *__restrict__ T, cx *P, const void *xbase, long long xoff, int N, bool wide,
                                           const float4 (&v)[(157 * SPS / 2 + 63) / 64], cx amp, float toa, int lane,
                                           float *sb, uint8_t *hb, cx *rawout, int nsoft) {
  typedef DemodGeom<SPS, NSMAX> G;
  constexpr int NLD = (157 * SPS / 2 + 63) / 64;
  const cx inv = cdiv(mk(1.0f, 0.0f), amp);                // ((complex)1.0)/channel (:1066)
  // delayVector(-TOA) bookkeeping (:577-582)
  const float delay = -toa;
  const int io = (int)floorf(delay);
  const float frac = delay - (float)io;
  const bool filt = fabs((double)frac) > 1e-2;
  float tp[21];
  {
    const float f512 = frac * 512.0f;
    const int f = (int)f512;
    // (frac = delay - floor(delay) can round to exactly 1.0 for a tiny negative delay: f = 512 is off the grid, the table
    // sinc below then forms sinc(pi (j - 10 - 1.0)) as the reference does)
    if (f < 512 && (float)f == f512) {                     // on the 1/512 grid: sinc_grid[f][j] (uniform -> s_load)
#pragma unroll
      for (int j = 0; j < 21; j++) tp[j] = T->sinc_grid[f][j];
    } else {
      const float tv = dev_sinc(T->sinT, TRX_PI_F * ((float)(lane - 10) - frac));   // :588
#pragma unroll
      for (int j = 0; j < 21; j++) tp[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tv), j));
    }
  }

  // ---- stage scaled samples at position n + io + C; zero the positions left uncovered ----
  const int lo = io + G::C, hi = N + io + G::C;            // samples occupy positions [lo, hi)
  for (int u = lane; u < lo && u < G::U; u += 64) P[(u % SPS) * G::QLEN + u / SPS] = mk(0, 0);
  for (int u = (hi > 0 ? hi : 0) + lane; u < G::U; u += 64) P[(u % SPS) * G::QLEN + u / SPS] = mk(0, 0);
  if (wide && lo >= 0 && (N & 1) == 0 && (SPS % 2) == 0) {
    // common case: nothing falls off the front; the tail past U is never read and is not written.
    // Lane's pair (2q, 2q+1), q = lane + 64 i, sits at positions
    // u0 = 2q + lo, u0 + 1; successive i move both by 128 positions = 128/SPS entries of the same phase.
    const int ua = 2 * lane + lo, ub = ua + 1;
    cx *pa = P + (ua % SPS) * G::QLEN + ua / SPS;
    cx *pb = P + (ub % SPS) * G::QLEN + ub / SPS;
#pragma unroll
    for (int i = 0; i < NLD; i++) {
      const int q = lane + 64 * i;
      if (q < N / 2) {
        if (ua + 128 * i < G::U) pa[i * (128 / SPS)] = cmul(mk(v[i].x, v[i].y), inv);   // scaleVector (:713-723)
        if (ub + 128 * i < G::U) pb[i * (128 / SPS)] = cmul(mk(v[i].z, v[i].w), inv);
      }
    }
  } else if (wide) {
#pragma unroll
    for (int i = 0; i < NLD; i++) {
      const int q = lane + 64 * i;
      if (q < N / 2) {
        const cx a = cmul(mk(v[i].x, v[i].y), inv), c = cmul(mk(v[i].z, v[i].w), inv);   // scaleVector (:713-723)
        const int u0 = 2 * q + lo, u1 = u0 + 1;
        if (u0 >= 0 && u0 < G::U) P[(u0 % SPS) * G::QLEN + u0 / SPS] = a;
        if (u1 >= 0 && u1 < G::U) P[(u1 % SPS) * G::QLEN + u1 / SPS] = c;
      }
    }
    if ((N & 1) && lane == 0) {
      const int u0 = N - 1 + lo;
      if (u0 >= 0 && u0 < G::U) P[(u0 % SPS) * G::QLEN + u0 / SPS] = cmul(SMP::ld(xbase, xoff + N - 1), inv);
    }
  } else {
    for (int n = lane; n < N; n += 64) {
      const int u0 = n + lo;
      if (u0 >= 0 && u0 < G::U) P[(u0 % SPS) * G::QLEN + u0 / SPS] = cmul(SMP::ld(xbase, xoff + n), inv);
    }
  }
  wave_lds_fence();

  const cx *rev = T->rev;
  for (int m0 = 0; m0 < nsoft; m0 += 64) {
    const int m = m0 + lane;
    const int t = SPS * m - io;                            // shifted[k] = filtered[k - intOffset] (:597-613)
    cx y = mk(0, 0);
    if (m < nsoft && t >= 0 && t < N) {
      if (filt) {
#pragma unroll
        for (int j = 0; j < 21; j++) {                     // convolve(...,NO_DELAY), 21 real taps, j ascending (:590)
          const int k = 10 - j + G::C;                     // position = SPS*m + k
          y = cadd(y, cmulr(P[(k % SPS) * G::QLEN + k / SPS + m], tp[j]));
        }
      } else {
        y = P[(G::C % SPS) * G::QLEN + G::C / SPS + m];
      }
    }
    if (RAW) {
      if (m < nsoft) rawout[m] = y;
    } else if (m < nsoft) {
      const cx rv = rev[SPS * m];
      const float re = rv.r * y.r - rv.i * y.i;            // real part of GMSKReverseRotate (:259-262)
      // vectorSlicer (:513-515): (float)(0.5*(double)(re + 1.0F)).  re + 1.0F is 0 or at least 2^-24 in
      // magnitude, so halving it is exact in float as well and the double round trip can go.
      float sv = (re + 1.0F) * 0.5F;
      if (sv > 1.0f) sv = 1.0f;
      if (sv < 0.0f) sv = 0.0f;
      sb[m] = sv;
      if (hb) hb[m] = sv > 0.5F;                           // SoftVector::bit (BitVector.h:415-420)
    }
  }
}


template <int SPS, int LPB>
struct FusedGeom {
  typedef DemodGeom<SPS, 148> D;
  static constexpr int NL = 36 * SPS;                      // correlation lags
  static constexpr int NE = 20 * SPS;                      // energyDetect window
  static constexpr int FRONT = 8 * SPS;                    // zero pad in front of the window
  static constexpr int CG = (NL + LPB - 1) / LPB;          // lags per lane: t = SPS*CG*g + p + SPS*i
  static constexpr int NSV = CG + 15;                      // window words a lane touches
  static constexpr int GA = (NL + SPS * CG - 1) / (SPS * CG);   // lane groups that own real lags
  static constexpr int WLEN = (GA * SPS * CG + 15 * SPS + 3) & ~3;
  static constexpr int PADC = 24;                          // zero pad either side of the correlation
  static constexpr int CLEN = NL + 2 * PADC;
  static constexpr int NV = 2 * (3 * SPS + 1);             // valley terms
  // scratch offsets in complex units (all even => 16-byte aligned)
  static constexpr int O_W = 0;
  static constexpr int O_C = O_W + WLEN;
  static constexpr int O_E = O_C + CLEN;
  static constexpr int O_LOC = O_E + NE / 2;
  static constexpr int O_V = O_LOC + 26;
  static constexpr int SCR = O_V + ((NV / 2 + 1) & ~1);
  static constexpr int REG = ((D::U > SCR ? D::U : SCR) + 1) & ~1;
  static constexpr int NLD = (157 * SPS / 2 + LPB - 1) / LPB;     // 16-byte sample pairs per lane
  static constexpr int OPL = (148 + LPB - 1) / LPB;               // soft bits per lane: m = OPL*hl + i
  static constexpr int BPW = 64 / LPB;                     // bursts per wave
};


// demodulateBurst (k_demod's arithmetic) for one detected burst whose samples sit in registers: pair
// q = hl + LPB*i of v[] holds samples 2q, 2q+1.  P: the burst's LDS staging area (DemodGeom<SPS,148>::U
// entries); whatever it held before is dead.  LPB lanes per burst; lane hl writes soft bits OPL*hl .. +OPL-1.
// staged(): called once the samples are in LDS and v[] is dead (k_normal_quad starts the next
// burst's loads there, into the same registers).
// tp_pre / rv_pre (optional): the 21 delay-filter taps for this TOA and the lane's OPL reverse-rotation
// values, when the caller has fetched them ahead of time.
// fused_demod_ex: the same with the staging step handed in -- stage(P, inv, lo) writes sample n, scaled by inv, to position
// u = n + lo (entry (u % SPS) * QLEN + u / SPS) for every n in [0, N) whose position lies in [0, U); the positions outside
// [lo, lo + N) have been zeroed.  fused_demod (below) stages from the 16-byte loads; k_demod_rx from samples it computes.
template <int SPS, int LPB, typename STAGE, typename HOOK>
__device__ __forceinline__ void fused_demod_ex(const TrxTables *__restrict__ T, cx *P, int N, cx amp, float toa, int hl, float *sb,
                                               uint8_t *hbp, int nsoft, STAGE stage, HOOK staged, const float *tp_pre, const cx *rv_pre) {
  typedef FusedGeom<SPS, LPB> G;
  typedef typename G::D D;
  const bool lane_owner = G::OPL * hl < 148;
  const int m0 = G::OPL * (lane_owner ? hl : 0);
  wave_lds_fence();                                        // scratch is dead: the staging area takes its place
  const cx inv = cdiv(mk(1.0f, 0.0f), amp);                // ((complex)1.0)/channel (:1066)
  const float delay = -toa;
  const int io = (int)floorf(delay);
  const float frac = delay - (float)io;
  const bool filt = fabs((double)frac) > 1e-2;
  float tp[21];
  if (tp_pre) {
#pragma unroll
    for (int j = 0; j < 21; j++) tp[j] = tp_pre[j];
  } else {
    const float f512 = frac * 512.0f;
    int f = (int)f512;
    if (f < 512 && (float)f == f512) {                     // on the 1/512 grid (always, after peakDetect; f = 512: see demod_core)
      if (LPB == 64) {                                     // wave-uniform: the row comes in by s_load
        f = __builtin_amdgcn_readfirstlane(f);
#pragma unroll
        for (int j = 0; j < 21; j++) tp[j] = T->sinc_grid[f & 511][j];
      } else {
        const float4 *row = reinterpret_cast<const float4 *>(T->sinc_grid[f & 511]);
#pragma unroll
        for (int q = 0; q < 6; q++) {
          const float4 r4 = row[q];
          if (4 * q < 21) tp[4 * q] = r4.x;
          if (4 * q + 1 < 21) tp[4 * q + 1] = r4.y;
          if (4 * q + 2 < 21) tp[4 * q + 2] = r4.z;
          if (4 * q + 3 < 21) tp[4 * q + 3] = r4.w;
        }
      }
    } else {                                               // (never after peakDetect; kept for completeness)
      const float tv = dev_sinc(T->sinT, TRX_PI_F * ((float)(hl - 10) - frac));   // :588, tap hl in lane hl
      const int first = (threadIdx.x & 63) & ~(LPB - 1);
#pragma unroll
      for (int j = 0; j < 21; j++) tp[j] = __shfl(tv, first + j, 64);
    }
  }
  const int lo = io + D::C, hi = N + io + D::C;            // samples occupy positions [lo, hi)
  for (int u = hl; u < lo && u < D::U; u += LPB) P[(u % SPS) * D::QLEN + u / SPS] = mk(0, 0);
  for (int u = (hi > 0 ? hi : 0) + hl; u < D::U; u += LPB) P[(u % SPS) * D::QLEN + u / SPS] = mk(0, 0);
  stage(P, inv, lo);
  wave_lds_fence();
  staged();

  const cx *rev = T->rev;
  cx y[G::OPL];
#pragma unroll
  for (int i = 0; i < G::OPL; i++) y[i] = mk(0, 0);
  if (filt) {
    // convolve(...,NO_DELAY), 21 real taps, j ascending (:590).  Output m0+i, tap j reads position
    // SPS*(m0+i) + c0 with c0 = 10 - j + C: the lane's OPL outputs share words, so walk the distinct
    // words c = c0 + SPS*i downwards (= j upwards for every output) and feed each to its outputs.
    constexpr int CMAX = 10 + D::C + SPS * (G::OPL - 1), CMIN = D::C - 10, NWD = CMAX - CMIN + 1;
#pragma unroll
    for (int w0 = 0; w0 < NWD; w0 += 8) {
      cx wd[8];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int c = CMAX - (w0 + q);
        if (c >= CMIN) wd[q] = P[(c % SPS) * D::QLEN + c / SPS + m0];
      }
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int c = CMAX - (w0 + q);
#pragma unroll
        for (int i = 0; i < G::OPL; i++) {
          const int j = 10 + D::C + SPS * i - c;
          if (c >= CMIN && j >= 0 && j <= 20) y[i] = cadd(y[i], cmulr(wd[q], tp[j]));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
#pragma unroll
    for (int i = 0; i < G::OPL; i++) y[i] = P[(D::C % SPS) * D::QLEN + D::C / SPS + m0 + i];
  }
#pragma unroll
  for (int i = 0; i < G::OPL; i++) {
    const int m = m0 + i;
    const int t = SPS * m - io;                            // shifted[k] = filtered[k - intOffset] (:597-613)
    if (lane_owner && m < nsoft) {
      const cx yy = (t >= 0 && t < N) ? y[i] : mk(0, 0);
      const cx rv = rv_pre ? rv_pre[i] : rev[SPS * m];
      const float re = rv.r * yy.r - rv.i * yy.i;          // real part of GMSKReverseRotate (:259-262)
      // vectorSlicer (:513-515): (float)(0.5*(double)(re + 1.0F)).  re + 1.0F is 0 or at least 2^-24 in
      // magnitude, so halving it is exact in float as well and the double round trip can go.
      float sv = (re + 1.0F) * 0.5F;
      if (sv > 1.0f) sv = 1.0f;
      if (sv < 0.0f) sv = 0.0f;
      sb[m] = sv;
      if (hbp) hbp[m] = sv > 0.5F;                         // SoftVector::bit (BitVector.h:415-420)
    }
  }
}


template <int SPS, int LPB, typename HOOK>
__device__ __forceinline__ void fused_demod(const TrxTables *__restrict__ T, cx *P, const float4 (&v)[(157 * SPS / 2 + LPB - 1) / LPB],
                                            int N, cx amp, float toa, int hl, float *sb, uint8_t *hbp, int nsoft, HOOK staged,
                                            const float *tp_pre, const cx *rv_pre) {
  typedef FusedGeom<SPS, LPB> G;
  typedef typename G::D D;
  auto stage = [&](cx *P_, cx inv, int lo) {
  if (lo >= 0 && (N & 1) == 0 && (2 * LPB) % SPS == 0) {
    // common case: nothing falls off the front, pairs are whole.  Pair q = hl + LPB*i sits at positions
    // u = 2q + lo, u + 1; successive i move both by 2*LPB positions = 2*LPB/SPS entries of the same phase.
    const int ua = 2 * hl + lo, ub = ua + 1;
    cx *pa = P_ + (ua % SPS) * D::QLEN + ua / SPS;
    cx *pb = P_ + (ub % SPS) * D::QLEN + ub / SPS;
#pragma unroll
    for (int i = 0; i < G::NLD; i++) {
      if (2 * (hl + LPB * i) < N) {
        if (ua + 2 * LPB * i < D::U) pa[i * (2 * LPB / SPS)] = cmul(mk(v[i].x, v[i].y), inv);   // scaleVector (:713-723)
        if (ub + 2 * LPB * i < D::U) pb[i * (2 * LPB / SPS)] = cmul(mk(v[i].z, v[i].w), inv);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < G::NLD; i++) {
      const int n0 = 2 * (hl + LPB * i);
      const int u0 = n0 + lo, u1 = u0 + 1;
      if (n0 < N && u0 >= 0 && u0 < D::U) P_[(u0 % SPS) * D::QLEN + u0 / SPS] = cmul(mk(v[i].x, v[i].y), inv);
      if (n0 + 1 < N && u1 >= 0 && u1 < D::U) P_[(u1 % SPS) * D::QLEN + u1 / SPS] = cmul(mk(v[i].z, v[i].w), inv);
    }
  }
  };
  fused_demod_ex<SPS, LPB>(T, P, N, amp, toa, hl, sb, hbp, nsoft, stage, staged, tp_pre, rv_pre);
}


// ---------------------------------------------------------------------------------------------
// The TOLERANCE-MODE demodulator (trxsig_set_soft_mode(ctx, TRXSIG_SOFT_TOLERANCE); the default stays the value-exact one).
// north_star grants "1e-4 relative on soft symbols, bit-exact on hard-decision bits"; detection, amplitude and TOA stay the
// reference's values bit for bit (they come in from k_tsc_peak2), only the arithmetic BEHIND them is rearranged:
//   * the samples are staged as they are -- scaleVector's 625 complex products (sigProcLib.cpp:713-723) go; 1/amp is folded
//     into the reverse rotation and applied to the 148 outputs: re = Re((rev[m] * inv) * Y), Y = sum_j tap[j] * x[...];
//   * the 21-tap delay filter (:584-590) accumulates with fused multiply-adds (one rounding per term instead of two);
//   * the slicer (:513-515) is one fma.
// ≈200 instead of ≈530 VALU instructions per burst.  Both forms round the same real number R = Re(rev * inv * sum tap * x)
// (same float32 inv, same taps, same samples):
//   reference order:  |re  - R| <= 26.2 u (|c|+|d|) S Z        u = 2^-24, (c,d) = rev[m], S = sum_j |tap[j]| <= 4 (every row of
//   this form:        |re' - R| <= 25.0 u (|c|+|d|) S Z        the sinc grid: tests/test_soft_tolerance.py), Z = max|x|_inf * (|inv.r|+|inv.i|)
// (scaleVector 2u per component, 22 u for the 21 rounded products and 21 rounded sums, 2u + propagation for the rotation; fma chain 21 u,
// the folded factor 2u, the last product-difference 2u), so |re - re'| <= 51.2 * 1.5 * 4 u Z < 308 u Z, and a soft bit (re + 1) / 2
// moves by at most 154 u Z + u = 9.2e-6 Z.  The fast form is only taken when Z <= 8 (and everything is far from the float
// range's ends), i.e. GUARANTEED |soft' - soft| <= 7.4e-5 on the [0, 1] scale; measured: <= 1.5e-6 (profiles/r05_parity_campaign.txt).
// HARD BITS ARE EXACT: a burst with a valid output whose |re'| is not above 512 u Z + 2^-22 (where the two forms could fall on
// different sides of the slicer's 0.5), or with a NaN anywhere (the test is written so that NaN fails it), or off the 1/512
// TOA grid, or with an odd geometry, is redone by the value-exact code in the same wave (the samples are still in registers):
// no list, no second launch.  Returns true when the burst's outputs have been stored.
// ---------------------------------------------------------------------------------------------
// (one asm statement per staged word: both components of every output it feeds; the marker is what tools/asm_stats.py and
//  tests/test_no_fma_contraction.py look for.  t*: wave-uniform taps in SGPRs -- at most one distinct SGPR per instruction.)
__device__ __forceinline__ void fma_tol_1(cx &y0, float t0, cx x) {
  asm("v_fma_f32 %0, %2, %4, %0 ; soft-tolerance\n\tv_fma_f32 %1, %3, %4, %1 ; soft-tolerance"
      : "+v"(y0.r), "+v"(y0.i) : "v"(x.r), "v"(x.i), "s"(t0));
}
__device__ __forceinline__ void fma_tol_2(cx &y0, float t0, cx &y1, float t1, cx x) {
  asm("v_fma_f32 %0, %4, %6, %0 ; soft-tolerance\n\tv_fma_f32 %1, %5, %6, %1 ; soft-tolerance\n\t"
      "v_fma_f32 %2, %4, %7, %2 ; soft-tolerance\n\tv_fma_f32 %3, %5, %7, %3 ; soft-tolerance"
      : "+v"(y0.r), "+v"(y0.i), "+v"(y1.r), "+v"(y1.i) : "v"(x.r), "v"(x.i), "s"(t0), "s"(t1));
}
__device__ __forceinline__ void fma_tol_3(cx &y0, float t0, cx &y1, float t1, cx &y2, float t2, cx x) {
  asm("v_fma_f32 %0, %6, %8, %0 ; soft-tolerance\n\tv_fma_f32 %1, %7, %8, %1 ; soft-tolerance\n\t"
      "v_fma_f32 %2, %6, %9, %2 ; soft-tolerance\n\tv_fma_f32 %3, %7, %9, %3 ; soft-tolerance\n\t"
      "v_fma_f32 %4, %6, %10, %4 ; soft-tolerance\n\tv_fma_f32 %5, %7, %10, %5 ; soft-tolerance"
      : "+v"(y0.r), "+v"(y0.i), "+v"(y1.r), "+v"(y1.i), "+v"(y2.r), "+v"(y2.i) : "v"(x.r), "v"(x.i), "s"(t0), "s"(t1), "s"(t2));
}
__device__ __forceinline__ float fma_tol(float a, float b, float c) {
  float r;
  asm("v_fma_f32 %0, %1, %2, %3 ; soft-tolerance" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// max(|a|, |b|, c) for c >= 0 (one instruction; a NaN operand is passed over -- see fused_demod_tol)
__device__ __forceinline__ float max3_abs(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// the larger of two NON-NEGATIVE floats as an integer maximum (same order; no canonicalising v_max x, x in front)
__device__ __forceinline__ float umax_f(float a, float b) {
  const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  return __uint_as_float(ua > ub ? ua : ub);
}
#define TRX_TOL_ZMAX 8.0f                     /* fast form only when max|x| * |1/amp|_1 <= this: |soft' - soft| <= 7.4e-5 guaranteed */
#define TRX_TOL_GUARD 3.0517578125e-05f       /* 512 u = 2^-15: |re'| must exceed this times Z (+ 2^-22) for the hard bit to be safe */

// fused_demod_tol_ex: the staging step handed in, as fused_demod_ex -- stage_raw(P, lo) writes sample n, AS IT IS, to position
// u = n + lo (entry (u % SPS) * QLEN + u / SPS) for every n in [0, N) whose position lies in [0, U) (lo may be negative: an access
// burst's delay shifts samples off the front), after the positions outside [lo, lo + N) have been zeroed.  xm_lane: max(|re|, |im|)
// over the samples this lane holds.
template <int SPS, typename STAGE>
__device__ __forceinline__ bool fused_demod_tol_ex(const TrxTables *__restrict__ T, cx *P, int N, cx amp, float toa, int hl, float *sb,
                                                   uint8_t *hbp, int nsoft, float xm_lane, STAGE stage_raw) {
  typedef FusedGeom<SPS, 64> G;
  typedef typename G::D D;
  static_assert(128 % SPS == 0, "pairs of a lane keep their phase from load to load");
  static_assert(G::OPL == 3, "three soft bits per lane");
  const bool lane_owner = G::OPL * hl < 148;
  const int m0 = G::OPL * (lane_owner ? hl : 0);
  const cx inv = cdiv(mk(1.0f, 0.0f), amp);                // ((complex)1.0)/channel (:1066): the reference's value
  const float delay = -toa;
  const int io = (int)floorf(delay);
  const float frac = delay - (float)io;
  const bool filt = fabs((double)frac) > 1e-2;
  const float f512 = frac * 512.0f;
  int f = (int)f512;
  const int lo = io + D::C, hi = N + io + D::C;            // samples occupy positions [lo, hi)
  // ---- may this burst take the fast form at all? (every quantity is wave-uniform) ----
  float xm = xm_lane;
  xm = umax_f(xm, dpp_f<0xB1>(xm));
  xm = umax_f(xm, dpp_f<0x4E>(xm));
  xm = umax_f(xm, dpp_f<0x141>(xm));
  xm = umax_f(xm, dpp_f<0x140>(xm));
  xm = umax_f(umax_f(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(xm), 0)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xm), 16))),
              umax_f(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(xm), 32)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xm), 48))));
  const float inv1 = fabsf(inv.r) + fabsf(inv.i);
  const float Z = xm * inv1;
  // (written so that a NaN or an infinity in amp / TOA fails; a NaN SAMPLE passes fmaxf unseen and is caught at the outputs below)
  const bool eligible = (f < 512) && ((float)f == f512) && (xm >= 1e-15f) && (xm <= 1e15f) && (inv1 >= 1e-15f) &&
                        (inv1 <= 1e15f) && (Z <= TRX_TOL_ZMAX);
  if (!__builtin_amdgcn_readfirstlane(eligible)) return false;
  f = __builtin_amdgcn_readfirstlane(f);
  float tp[21];
#pragma unroll
  for (int j = 0; j < 21; j++) tp[j] = T->sinc_grid[f & 511][j];      // wave-uniform: s_load

  // ---- stage the samples AS THEY ARE at position n + io + C (polyphase order); zero the positions left uncovered ----
  wave_lds_fence();
  for (int u = hl; u < lo && u < D::U; u += 64) P[(u % SPS) * D::QLEN + u / SPS] = mk(0, 0);
  for (int u = (hi > 0 ? hi : 0) + hl; u < D::U; u += 64) P[(u % SPS) * D::QLEN + u / SPS] = mk(0, 0);
  stage_raw(P, lo);
  wave_lds_fence();

  cx y[G::OPL];
#pragma unroll
  for (int i = 0; i < G::OPL; i++) y[i] = mk(0, 0);
  if (filt) {
    // the word walk of fused_demod_ex: output m0+i, tap j reads position SPS*(m0+i) + c0, c0 = 10 - j + C
    constexpr int CMAX = 10 + D::C + SPS * (G::OPL - 1), CMIN = D::C - 10, NWD = CMAX - CMIN + 1;
#pragma unroll
    for (int w0 = 0; w0 < NWD; w0 += 8) {
      cx wd[8];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int c = CMAX - (w0 + q);
        if (c >= CMIN) wd[q] = P[(c % SPS) * D::QLEN + c / SPS + m0];
      }
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int c = CMAX - (w0 + q);
        const int j0 = 10 + D::C - c, j1 = j0 + SPS, j2 = j1 + SPS;
        const bool u0 = c >= CMIN && j0 >= 0 && j0 <= 20, u1 = c >= CMIN && j1 >= 0 && j1 <= 20, u2 = c >= CMIN && j2 >= 0 && j2 <= 20;
        const float t0 = tp[u0 ? j0 : 0], t1 = tp[u1 ? j1 : 0], t2 = tp[u2 ? j2 : 0];
        if (u0 && u1 && u2) fma_tol_3(y[0], t0, y[1], t1, y[2], t2, wd[q]);
        else if (u0 && u1) fma_tol_2(y[0], t0, y[1], t1, wd[q]);
        else if (u1 && u2) fma_tol_2(y[1], t1, y[2], t2, wd[q]);
        else if (u0 && u2) fma_tol_2(y[0], t0, y[2], t2, wd[q]);
        else if (u0) fma_tol_1(y[0], t0, wd[q]);
        else if (u1) fma_tol_1(y[1], t1, wd[q]);
        else if (u2) fma_tol_1(y[2], t2, wd[q]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
#pragma unroll
    for (int i = 0; i < G::OPL; i++) y[i] = P[(D::C % SPS) * D::QLEN + D::C / SPS + m0 + i];
  }
  const cx *rev = T->rev;
  const float guard = fma_tol(Z, TRX_TOL_GUARD, 2.384185791015625e-07f);
  float sv[G::OPL];
  bool risky = false;
#pragma unroll
  for (int i = 0; i < G::OPL; i++) {
    const int m = m0 + i;
    const int t = SPS * m - io;                            // shifted[k] = filtered[k - intOffset] (:597-613)
    const cx rv = rev[SPS * m];
    // (rev[m] * inv) * Y, real part: the reference rounds x * inv, the sums and rev * y one by one (see the bound above)
    const float a = fma_tol(rv.r, inv.r, -(rv.i * inv.i));
    const float b = fma_tol(rv.r, inv.i, rv.i * inv.r);
    const float re = fma_tol(a, y[i].r, -(b * y[i].i));
    const bool in_range = (t >= 0 && t < N);               // else the reference's filtered sample is 0: soft 0.5, hard 0, either form
    float s = fma_tol(re, 0.5F, 0.5F);                     // vectorSlicer (:513-515): (re + 1) / 2, halving is exact
    s = fminf(fmaxf(s, 0.0f), 1.0f);
    sv[i] = in_range ? s : 0.5F;
    risky = risky || (lane_owner && m < nsoft && in_range && !(fabsf(re) > guard));
  }
  if (__builtin_amdgcn_ballot_w64(risky) != 0) return false;          // the value-exact form redoes this burst
#pragma unroll
  for (int i = 0; i < G::OPL; i++) {
    const int m = m0 + i;
    if (lane_owner && m < nsoft) {
      sb[m] = sv[i];
      if (hbp) hbp[m] = sv[i] > 0.5F;                      // SoftVector::bit (BitVector.h:415-420)
    }
  }
  return true;
}


// the same from the 16-byte loads of k_demod / k_normal_quad / k_normal_chain: pair q = hl + 64 i of v[] holds samples 2q, 2q+1
template <int SPS>
__device__ __forceinline__ bool fused_demod_tol(const TrxTables *__restrict__ T, cx *P, const float4 (&v)[(157 * SPS / 2 + 63) / 64],
                                                int N, cx amp, float toa, int hl, float *sb, uint8_t *hbp, int nsoft) {
  typedef FusedGeom<SPS, 64> G;
  typedef typename G::D D;
  float xm = 0.0f;
#pragma unroll
  for (int i = 0; i < G::NLD; i++) {                       // (lanes past the burst's end hold zeros)
    xm = max3_abs(v[i].x, v[i].y, xm);
    xm = max3_abs(v[i].z, v[i].w, xm);
  }
  auto stage_raw = [&](cx *P_, int lo) {
    if (lo >= 0) {                                          // the common case: nothing falls off the front
      const int ua = 2 * hl + lo, ub = ua + 1;
      cx *pa = P_ + (ua % SPS) * D::QLEN + ua / SPS;
      cx *pb = P_ + (ub % SPS) * D::QLEN + ub / SPS;
#pragma unroll
      for (int i = 0; i < G::NLD; i++) {
        if (2 * (hl + 64 * i) < N) {
          if (ua + 128 * i < D::U) pa[i * (128 / SPS)] = mk(v[i].x, v[i].y);
          if (ub + 128 * i < D::U) pb[i * (128 / SPS)] = mk(v[i].z, v[i].w);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < G::NLD; i++) {
        const int n0 = 2 * (hl + 64 * i);
        const int u0 = n0 + lo, u1 = u0 + 1;
        if (n0 < N && u0 >= 0 && u0 < D::U) P_[(u0 % SPS) * D::QLEN + u0 / SPS] = mk(v[i].x, v[i].y);
        if (n0 + 1 < N && u1 >= 0 && u1 < D::U) P_[(u1 % SPS) * D::QLEN + u1 / SPS] = mk(v[i].z, v[i].w);
      }
    }
  };
  return fused_demod_tol_ex<SPS>(T, P, N, amp, toa, hl, sb, hbp, nsoft, xm, stage_raw);
}

template <int SPS, bool RAW, int NSMAX, typename SMP = SmpC32, bool TOL = false>
__global__ __launch_bounds__(64 * TRX_DEMOD_WAVES) void k_demod(const TrxTables *__restrict__ T,
                                               const void *__restrict__ samples,
                                               const int32_t *__restrict__ offset,
                                               const int32_t *__restrict__ length, int B,
                                               const cx *__restrict__ amp_in,
                                               const float *__restrict__ toa_in,
                                               const uint8_t *__restrict__ flags, int need_mask,
                                               float *__restrict__ soft, uint8_t *__restrict__ hard,
                                               int nsoft, int stride) {
  // RAW: `soft` is really a complex array (stride complex per burst) that receives the delayed,
  // scaled burst itself (every sample, no rotation/slicing): the delayVector step of equalizeBurst.
  typedef DemodGeom<SPS, NSMAX> G;
  __shared__ cx ph[TRX_DEMOD_WAVES][G::U];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x * TRX_DEMOD_WAVES + wave;       // wave-uniform
  if (b >= B) return;
  float *sb = soft + (size_t)b * stride;
  uint8_t *hb = hard ? hard + (size_t)b * stride : nullptr;
  const int off = offset[b], N = length[b];
  const cx amp = amp_in[b];
  const float toa = toa_in[b];
  bool enabled = (off >= 0) && (N >= 92 * SPS) && (N <= 157 * SPS) && (N % SPS == 0) &&
                 (fabsf(toa) <= 4096.0f);                  // also rejects NaN/inf TOA
  if (flags) enabled = enabled && (need_mask ? ((flags[b] & need_mask) == need_mask) : (flags[b] != 0));
  if (RAW) nsoft = enabled ? N : 0;
  if (!enabled) {
    if (!RAW) for (int m = lane; m < nsoft; m += 64) { sb[m] = 0.0f; if (hb) hb[m] = 0; }
    return;
  }
  // ---- issue the burst's loads first (two samples per lane per load: 16 bytes of float32 pairs, 8 of fp16 pairs) ----
  constexpr int NLD = (157 * SPS / 2 + 63) / 64;           // pair loads per lane
  const bool wide = (off & 1) == 0;
  float4 v[NLD];
  if (wide) {
#pragma unroll
    for (int i = 0; i < NLD; i++) {
      const int q = lane + 64 * i;
      v[i] = (q < N / 2) ? SMP::ld2(samples, off, q) : make_float4(0, 0, 0, 0);
    }
  }
  // the common case (148 soft bits, even offset and length) goes through fused_demod: same arithmetic,
  // but a lane owns three CONSECUTIVE soft bits, whose filter windows share 34 of their 63 staged words
  // (measured: 68.0 -> 64.6 us per 64 K bursts)
  if (!RAW && NSMAX == 148 && wide && (N & 1) == 0) {
    if (TOL) {                                             // tolerance mode: the rearranged form, unless this burst has to be exact
      if (fused_demod_tol<SPS>(T, ph[wave], v, N, amp, toa, lane, sb, hb, nsoft)) return;
    }
    fused_demod<SPS, 64>(T, ph[wave], v, N, amp, toa, lane, sb, hb, nsoft, [] {}, nullptr, nullptr);
    return;
  }
  demod_core<SPS, RAW, NSMAX, SMP>(T, ph[wave], samples, off, N, wide, v, amp, toa, lane, sb, hb,
                              RAW ? reinterpret_cast<cx *>(soft) + (size_t)b * stride : nullptr, nsoft);
}


}  // namespace
