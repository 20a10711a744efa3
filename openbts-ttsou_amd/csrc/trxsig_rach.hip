// trxsig_rach.hip -- the access-burst path: detectRACHBurst (sigProcLib.cpp:860-914) as k_rach_fast
// (approximate steering + exact recomputation) or k_rach_corr / k_rach_peak (exact at every lag).
// Numerical contract: see trxsig_dev.h / DESIGN.md (every float32 operation is the reference's, in the
// reference's order; built with -ffp-contract=off).
#include "trxsig_bisect.h"
#include "trxsig_corr.h"      // energy_chain (the DPP row-shift energy sum)
#include "trxsig_rxgen.h"     // bursts computed from the raw int16 stream (k_rach_front_rx)

namespace {

// ---------------------------------------------------------------------------------------------
// quad_bisect: peakDetect's bisection with FOUR lanes per burst: lanes {0,1,2,3} of a quad own the
// four independent 21-term chains of a step (early.re, early.im, late.re, late.im); |.|^2 and the
// early/late comparison are exchanged inside the quad with DPP quad_perm, and the sinc row is shared
// by the quad (each lane keeps a quarter, taps are broadcast by quad_perm).  Same arithmetic as
// peak_bisect.  Used where one wave owns one burst (k_rach_fast).
// ---------------------------------------------------------------------------------------------

template <int CTRL>
__device__ __forceinline__ float quad_perm(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// A sinc_grid row (24 floats) is shared by the 4 lanes of a quad: lane q keeps float4 #q and #(q+4)
// (8 floats), tap j is broadcast from lane (j/4)%4 with a DPP quad_perm.
struct QRow { float4 a, b; };

template <int J>
__device__ __forceinline__ float qrow_tap(const QRow &w) {
  constexpr int f4 = J / 4, src = f4 % 4, comp = J % 4;
  const float4 &v = (f4 < 4) ? w.a : w.b;
  const float x = comp == 0 ? v.x : (comp == 1 ? v.y : (comp == 2 ? v.z : v.w));
  return quad_perm<src * 0x55>(x);                         // quad_perm:[src,src,src,src]
}

template <bool MASK, int STR, int J>
__device__ __forceinline__ float qchain(const float *p, int slot0, int zslot, const QRow &w, float acc) {
  if constexpr (J < 21) {
    float v = p[J * 2 * STR];
    if (MASK && slot0 + J > zslot) v = 0.0f;               // interpolatePoint never uses the last sample (:646)
    acc = acc + v * qrow_tap<J>(w);
    return qchain<MASK, STR, J + 1>(p, slot0, zslot, w, acc);
  } else {
    return acc;
  }
}

// HOFF: slot of lag M-12; STR: complex entries per slot row (bursts side by side)
template <int HOFF, int STR, bool MASK>
__device__ __forceinline__ void quad_bisect(const TrxTables *__restrict__ T, const float *rcf, int bi, int q, int M,
                                            int zslot, float *peakIx, float *pk_own, float *pk_partner) {
  const int c = q & 1, late = q >> 1;
  auto load_row = [&](int f) {
    const float4 *row = reinterpret_cast<const float4 *>(T->sinc_grid[f & 511]);
    QRow w;
    w.a = row[q];
    w.b = (q < 2) ? row[q + 4] : make_float4(0, 0, 0, 0);
    return w;
  };
  // one chain of interpolatePoint: sum_j comp(corr[slot0 + j]) * s[j], j ascending
  auto chain = [&](int slot0, const QRow &w) {
    const float *p = rcf + ((size_t)slot0 * STR + bi) * 2 + c;
    return qchain<MASK, STR, 0>(p, slot0, zslot, w, 0.0f);
  };
  auto frac512 = [](float ix) { return (int)((ix - floorf(ix)) * 512.0f); };
  auto slot_of = [&](float ix) {
    int base = (int)floorf(ix) - M + 2;                    // 0..3 by construction
    base = base < 0 ? 0 : (base > 3 ? 3 : base);
    return base + HOFF;
  };

  float early = (float)M - 1;
  float incr = 0.5f;
  bool active = true;
  QRow cur = load_row(0);
#pragma unroll 1
  for (int step = 0; step < 9; step++) {
    const QRow up = load_row(frac512(early + incr));
    const QRow dn = load_row(frac512(early - incr));
    const float a = chain(slot_of(early) + 2 * late, cur);
    const float sq = a * a;
    const float osq = quad_perm<0xB1>(sq);                 // partner component: lanes 0<->1, 2<->3
    const float nrm = c ? (sq + osq) : (osq + sq);         // i*i + r*r (Complex.h:119)
    const float onrm = quad_perm<0x4E>(nrm);               // the other point: lanes 0,1 <-> 2,3
    const float ne = late ? onrm : nrm, nl = late ? nrm : onrm;
    const bool goUp = ne < nl, goDn = ne > nl;
    if (active) {
      if (goUp) early += incr;
      else if (goDn) early -= incr;
      else active = false;                                 // "else break" (:695)
      if (active) incr = incr * 0.5f;
    }
    if (active) cur = goUp ? up : dn;                      // the row changes only if the index moved
  }
  *peakIx = early + 1.0f;
  const float a = chain(slot_of(*peakIx), cur);            // every lane: its own component of the peak
  *pk_own = a;
  *pk_partner = quad_perm<0xB1>(a);
}

// ---------------------------------------------------------------------------------------------
// k_rach_corr: detectRACHBurst's correlation (sigProcLib.cpp:867-869) over ALL lags with the dense
//   41*sps-tap access-burst sequence, energyDetect, argmax and the record for k_rach_peak.
//   One wave per burst; lane handles lags t = lane + 64*c.
//
// corr[t] = sum_j tmp[j]*x[t+s-j], tmp = reverse(conj(rach)), i.e. sum_{m=Lb-1..0} x[t-F+m]*conj(rach[m])
// with F = Lb/2, accumulated in that order (ascending j) -- every term of the reference, nothing
// factored or reordered, so corr is bit-identical; out-of-range samples are zeros in the padded LDS
// copy instead of being skipped (adds +-0).
//
// Record per burst (SoA, [slot][Bpad]): complex slots 0..23 = corr[M-12..M+11], slot 24 = {M, energy};
// then NVAL float slots = |corr|^2 at lags M-1+57*sps .. M+1+107*sps for the valley sum (:888-893).
// ---------------------------------------------------------------------------------------------
template <int SPS>
struct RachGeom {
  static constexpr int LB = 41 * SPS;                      // taps
  static constexpr int F = LB / 2;                         // front pad
  static constexpr int NMAX = 157 * SPS;
  static constexpr int NCL = (NMAX + 63) / 64;             // lags per lane
  static constexpr int XPAD = 64 * NCL + LB;               // padded burst length
  static constexpr int V0 = 57 * SPS - 1, V1 = 107 * SPS + 1;   // valley lags relative to M
  static constexpr int NVAL = V1 - V0 + 1;
  static constexpr int NE = 20 * SPS;
  static constexpr int NEQ = (NE + 15) / 16;
  static constexpr int CSLOTS = 25;                        // complex slots
};

#ifdef TRX_TUNING_BUILD   /* the exact-at-every-lag route (TRXSIG_TUNE_RACH_PATH 0): A/B reference only, not in the product library */
template <int SPS>
__global__ __launch_bounds__(256) void k_rach_corr(const TrxTables *__restrict__ T,
                                                   const cx *__restrict__ samples,
                                                   const int32_t *__restrict__ offset,
                                                   const int32_t *__restrict__ length, int B,
                                                   cx *__restrict__ rec, float *__restrict__ recv, int Bpad) {
  typedef RachGeom<SPS> G;
  __shared__ cx xs[4][G::XPAD];                            // zero-padded burst, later reused for corr
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  const int off = offset[b], N = length[b];
  const bool good = (off >= 0) && (N >= 92 * SPS) && (N <= 157 * SPS) && (N % SPS == 0);
  const cx *x = samples + (good ? off : 0);
  cx *X = xs[wave];

  for (int i = lane; i < G::XPAD; i += 64) {
    const int n = i - G::F;
    X[i] = (good && n >= 0 && n < N) ? x[n] : mk(0, 0);
  }
  // energyDetect on the first 20*sps samples, strictly in order (row 0 of the wave does the chain)
  float nrm[G::NEQ];
#pragma unroll
  for (int q = 0; q < G::NEQ; q++) {
    const int i = (lane & 15) + 16 * q;
    cx v = mk(0, 0);
    if (good && i < G::NE) v = x[i];
    nrm[q] = norm2(v);
  }
  float energy = energy_chain<SPS, 0>(0.0f, nrm);
  energy = __shfl(energy, 0, 64);
  wave_lds_fence();

  cx acc[G::NCL];
#pragma unroll
  for (int c = 0; c < G::NCL; c++) acc[c] = mk(0, 0);
  const cx *rseq = T->rach;
#pragma unroll 4
  for (int m = G::LB - 1; m >= 0; m--) {
    const cx rm = rseq[m];
    const cx tp = mk(rm.r, -rm.i);                         // conj (:487)
#pragma unroll
    for (int c = 0; c < G::NCL; c++) acc[c] = cadd(acc[c], cmul(X[lane + 64 * c + m], tp));
  }
  wave_lds_fence();                                        // all lanes are done with the samples

  float bestP = 0.0f;
  int bestT = -1;
#pragma unroll
  for (int c = 0; c < G::NCL; c++) {
    const int t = lane + 64 * c;
    if (t < N && good) {
      X[t] = acc[c];
      const float p = norm2(acc[c]);
      if (p > bestP) { bestP = p; bestT = t; }             // strict >, first maximum (:675)
    }
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) {
    const float oP = __shfl_xor(bestP, m, 64);
    const int oT = __shfl_xor(bestT, m, 64);
    const bool take = (oP > bestP) || (oP == bestP && oT >= 0 && (bestT < 0 || oT < bestT));
    if (take) { bestP = oP; bestT = oT; }
  }
  wave_lds_fence();

  const int M = bestT;
  if (lane < G::CSLOTS) {
    cx v = mk(0, 0);
    if (lane < 24) {
      const int lag = M - 12 + lane;
      if (good && lag >= 0 && lag < N) v = X[lag];
    } else {
      v = mk(__int_as_float(good ? M : -2), energy);
    }
    rec[(size_t)lane * Bpad + b] = v;
  }
  for (int s = lane; s < G::NVAL; s += 64) {
    const int lag = M + G::V0 + s;
    float p = 0.0f;
    if (good && lag >= 0 && lag < N) p = norm2(X[lag]);
    recv[(size_t)s * Bpad + b] = p;
  }
}

// k_rach_peak: one lane per burst: peakDetect bisection, bogus-TOA check, valley RMS over
//   peak+57*sps .. peak+107*sps, threshold, amp = peak/gain, TOA bookkeeping (sigProcLib.cpp:873-913).
template <int SPS>
__global__ __launch_bounds__(64) void k_rach_peak(const TrxTables *__restrict__ T,
                                                  const cx *__restrict__ rec, const float *__restrict__ recv,
                                                  const int32_t *__restrict__ length, int Bpad, int B,
                                                  float detect_thresh, float energy_thresh,
                                                  uint8_t *__restrict__ flags, cx *__restrict__ amp_out,
                                                  float *__restrict__ toa_out,
                                                  float *__restrict__ avgpwr_out) {
  typedef RachGeom<SPS> G;
  __shared__ cx loc[26][64];
  const int lane = threadIdx.x;
  const int b = blockIdx.x * 64 + lane;
  const bool live = b < B;
  const int bb = live ? b : B - 1;
  const cx meta = rec[(size_t)24 * Bpad + bb];
  const int M = __float_as_int(meta.r);
  const float energy = meta.i;
  const bool good = M != -2;
  const int N = length[bb];
#pragma unroll
  for (int j = 0; j < 24; j++) {
    const cx v = rec[(size_t)j * Bpad + bb];
    const int lag = M - 12 + j;
    loc[j][lane] = (lag > N - 2) ? mk(0, 0) : v;           // interpolatePoint never uses the last sample (:646)
  }
  loc[24][lane] = mk(0, 0);
  loc[25][lane] = mk(0, 0);

  float peakIx;
  const cx peak = peak_bisect<64>(T->sinc_grid, loc, lane, M, &peakIx);

  float toa = peakIx;
  cx amp = mk(0, 0);
  bool detected = false;
  const bool energy_ok = good && (energy_thresh < 0.0f ||
                                  energy / (float)(unsigned)G::NE > energy_thresh * energy_thresh);
  if (!(toa < 0.0f) && !(toa > (float)N) && good) {        // :878-882
    const int p = (int)rintf(toa);
    float valley = 0.0f, numSamples = 0.0f;
#pragma unroll 4
    for (int i = 57 * SPS; i <= 107 * SPS; i++) {          // :888-893, this order, stop at the end
      const int lag = p + i;
      int sl = lag - M - G::V0;                            // 0 .. NVAL-1 because |p - M| <= 1
      sl = sl < 0 ? 0 : (sl > G::NVAL - 1 ? G::NVAL - 1 : sl);
      const float v = recv[(size_t)sl * Bpad + bb];
      if (lag < N) { valley += v; numSamples += 1.0f; }
    }
    if (numSamples >= 2) {
      const float RMS = (float)((double)sqrtf(valley / numSamples) + 0.00001);      // :901
      const float peakToMean = sqrtf(norm2(peak)) / RMS;
      amp = cdiv(peak, T->rach_gain);                      // :905
      toa = toa - T->rach_toa - (float)(8 * SPS);          // :907
      detected = peakToMean > detect_thresh;
    }
  }
  if (!energy_ok) { amp = mk(0, 0); toa = 0.0f; detected = false; }

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

#endif  // TRX_TUNING_BUILD

// ---------------------------------------------------------------------------------------------
// k_rach_fast: detectRACHBurst (sigProcLib.cpp:860-914) with the SAME results as k_rach_corr +
//   k_rach_peak at a fraction of the arithmetic.  One wave per burst.
//
// Only three things in detectRACHBurst depend on exact correlation values: which lag is the
// maximum, the 24 lags around it that peakDetect interpolates, and (through one comparison) the
// valley power.  So:
//  1. an APPROXIMATE correlation at all lags: the access-burst sequence is GMSK-modulated symbols,
//     rach[m] = sum_k p[m+sps-sps*k] * c_k, hence corr[t] = sum_k conj(c_k) * z[t-F-sps+sps*k] with
//     z = x filtered by the (2*sps+1)-tap pulse, minus two edge terms for the pulse tails the
//     reference's NO_DELAY convolution dropped (sigProcLib.cpp:559); with c_k ~ (+-1)*i^k the 41-tap sum
//     is additions only.  Measured error of |corr|^2 <= 1.2e-5 of the maximum (tools/).
//  2. every lag whose approximate power is within RACH_DELTA (4e-3, >300x that error) of the
//     approximate maximum, plus the 26 lags around the approximate argmax, is recomputed EXACTLY (the
//     reference's 41*sps-term sum in its order, one lag per lane); the exact first-maximum among them
//     is the reference's argmax because no other lag can reach it.  If the exact argmax moved by more
//     than one lag its neighbourhood is recomputed too.
//  3. peakDetect's bisection runs on the exact neighbourhood (four lanes, quad_bisect).
//  4. the valley RMS uses the approximate powers with an error bar (rach_decide below); only if the threshold falls
//     inside the bar are the valley lags recomputed exactly and summed in the reference's order, so the detect
//     decision is the reference's in every case, for any threshold.
//
// Error model (what RACH_DELTA alone and the old fixed 1e-3 guard did not cover: large thresholds, bursts whose
// correlation maximum is small against their energy).  Let e[t] be the reference's float value of corr[t], a[t] the
// approximate pass's.  In exact arithmetic a[t] = sum_m conj(bi[m]) x[t-F+m] with the IMPLIED sequence
// bi[m] = sum_{sps k + j - sps = m} sym_k i^k p[j] (the edge terms remove m < 0 and m >= LB), while the reference
// uses the table sequence b[m] (rotation table values instead of i^k, float roundings of modulateBurst).  Hence
//   |a[t] - e[t]| <= |db|_2 |x|_2  +  (ga + ge) sum_m babs[m] |x[t-F+m]|  <=  (|db|_2 + 516 u |babs|_2) sqrt(Ex)  =: D
// with db = b - bi, babs[m] = sum p[j] >= |b[m]|, |bi[m]| (|babs|_2 <= sqrt2 |b|_2: neighbouring symbols are in
// quadrature), Ex = sum |x[n]|^2 over the burst, u = 2^-24; ge <= sqrt2 (164 + 2) u bounds the float error of the
// reference's 164-term sequential sum (two rounded products and a rounded add per term and component), ga <=
// sqrt2 (9 + 41 + 6) u that of the pulse filter (9 fused steps), the 41 additions and the edge terms; 516 u is more
// than their sum.  trx_rach_amp_err() evaluates |db|_2 + 516 u |babs|_2 on the host in double from the uploaded
// tables (so it follows the tables, whatever they are) and the kernels get it as an argument; D = that x sqrt(Ex).
//   * contenders: lag t can hold the reference's maximum only if sqrt(pw[t]) >= sqrt(pmax) - 2 D (pw: approximate
//     powers); the cut is the looser of that and RACH_DELTA.  When D is so large that the cut is void every lag
//     contends and the burst takes the exact route.
//   * valley: |sum_i pw[p+i] - sum_i |e[p+i]|^2| <= 2 D sqrt(cnt vs) + cnt D^2 (Cauchy-Schwarz), plus 3e-5 (vs + that)
//     for the float summations on both sides (cnt <= 201 terms) and the rounding of each norm.  peak/RMS is a
//     monotone function of the valley sum, also in float (division, sqrt, and the additions are monotone when
//     correctly rounded), so the reference's value lies between the values at vs + E and vs - E: above the threshold
//     at vs + E means detected, not above it at vs - E means not detected, anything else is recomputed exactly.
// If more far-away candidates turn up than fit in one pass (flat noise, silence) the burst takes
// the exact route for all lags.
// ---------------------------------------------------------------------------------------------
#ifndef TRX_RACH_EXACT_GROUP
#define TRX_RACH_EXACT_GROUP 4   /* taps per software-pipeline stage of rach_exact_lag */
#endif
#define RACH_DELTA 4e-3f                                    // contenders: at least everything within 0.4 % of the approximate maximum

// detectRACHBurst's decision (sigProcLib.cpp:901-903) from an approximate valley sum vs (cnt terms) whose lags'
// amplitudes are each within dlt of the reference's: 1 = detected, 0 = not, -1 = cannot be told from vs (see above)
__device__ __forceinline__ int rach_decide(float peak_abs, float vs, int cnt, float dlt, float thresh) {
  const float fc = (float)cnt;
  const float e1 = 2.0f * dlt * sqrtf(fc * vs) + fc * dlt * dlt;
  const float e = (e1 + 3e-5f * (vs + e1)) * 1.001f;
  if (!(vs == vs) || !(e == e) || !(vs >= 0.0f)) return -1;
  const float vhi = vs + e, vlo = vs > e ? vs - e : 0.0f;
  const float ptm_lo = peak_abs / (float)((double)sqrtf(vhi / fc) + 0.00001);   // :901-902 at the largest valley
  const float ptm_hi = peak_abs / (float)((double)sqrtf(vlo / fc) + 0.00001);   // ... and at the smallest
  if (ptm_lo > thresh) return 1;
  if (!(ptm_hi > thresh)) return 0;
  return -1;
}
struct RachSym {                                           // 2*bit-1 of gRACHSynchSequence (GSM/GSMCommon.cpp:57)
  static constexpr signed char v[41] = {
    -1, 1, -1, -1, 1, -1, 1, 1, -1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, -1, 1, -1, -1, -1, 1, 1, 1, 1, -1,
    -1, -1 };
};

template <int SPS>
struct RachFast {
  typedef RachGeom<SPS> R;
  static constexpr int XF = R::F + SPS;                     // X[i] = x[i - XF]
  static constexpr int XPAD = 64 * R::NCL + R::LB + 4 * SPS + 8;
  static constexpr int ZPAD = 64 * R::NCL + 40 * SPS + 2;   // Zs[i] = sum_j p[j] X[i+j]; even: written in pairs
  static constexpr int NB = 26;                             // lags M~-13 .. M~+12 always recomputed
};

// exact corr[t] (sigProcLib.cpp:474-503 + 322-366): sum over m = LB-1 .. 0 of x[t-F+m]*conj(rach[m])
template <int SPS>
__device__ __forceinline__ cx rach_exact_lag(const cx *X, const cx *__restrict__ rseq, int t) {
  typedef RachGeom<SPS> R;
  cx acc = mk(0, 0);
  const cx *xp = X + t + SPS;                              // X index of x[t-F+m] is t + m + SPS
  // Neighbouring samples are read through two bases the compiler cannot relate (xo = xp + an opaque zero): it would otherwise
  // pair them into ds_read2_b64, which the LDS serves at half the rate of two ds_read_b64 (4 array cycles per sample instead
  // of 2; the file is compiled without the backend's load/store merging for the same reason -- see the Makefile).
  int opaque0 = 0;
  asm volatile("" : "+s"(opaque0));
  const cx *xo = xp + opaque0;
  // Software pipelined in groups of TRX_RACH_EXACT_GROUP taps: the next group's samples (LDS) and taps (scalar loads) are in flight
  // while this group's 32 VALU run -- the wave shares its SIMD with only one or two others, so an exposed LDS/scalar
  // latency per group (the plain loop: 41 x ~250 cycles) is not hidden by anybody else.  Order of the sum unchanged.
  constexpr int GS = TRX_RACH_EXACT_GROUP, REM = R::LB % GS, NG = R::LB / GS;
  int m = R::LB - 1;
#pragma unroll
  for (int q = 0; q < REM; q++, m--) {
    const cx rm = rseq[m];
    acc = cadd(acc, cmul(xp[m], mk(rm.r, -rm.i)));
  }
  cx xa[GS], ta[GS];
#pragma unroll
  for (int q = 0; q < GS; q++) { xa[q] = (q & 1) ? xo[m - q] : xp[m - q]; ta[q] = rseq[m - q]; }
#pragma unroll 2
  for (int g = 0; g < NG; g++) {
    cx xb[GS], tb[GS];
    const int mn = (g + 1 < NG) ? m - GS : m;              // (the last iteration re-reads its own group: in range, unused)
#pragma unroll
    for (int q = 0; q < GS; q++) { xb[q] = (q & 1) ? xo[mn - q] : xp[mn - q]; tb[q] = rseq[mn - q]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < GS; q++) acc = cadd(acc, cmul(xa[q], mk(ta[q].r, -ta[q].i)));
    asm volatile("" : "+v"(acc.r), "+v"(acc.i));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < GS; q++) { xa[q] = xb[q]; ta[q] = tb[q]; }
    m = mn;
  }
  return acc;
}

// Approximate correlation, additions only: conj(c_k) z with c_k = sym_k i^k.  Lag t = lane + 64c needs
// Z[t + sps k] = Z[lane + sps (CJ c + k)], CJ = 64/sps: one read of Z[lane + sps j] serves every (c, k) with CJ c + k = j,
// and every accumulator still receives its terms k ascending.  (Template recursion: the (j, c) -> k map is resolved at
// compile time; a 185 x 10 loop nest is beyond what "#pragma unroll" unrolls.)
template <int SPS, int NCL, int J, int C>
__device__ __forceinline__ void rach_steer_acc(const cx z, float (&ar)[NCL], float (&ai)[NCL]) {
  if constexpr (C < NCL) {
    constexpr int k = J - (64 / SPS) * C;
    if constexpr (k >= 0 && k <= 40) {
      constexpr float sg = (float)RachSym::v[k];
      if constexpr ((k & 3) == 0) { ar[C] += sg * z.r; ai[C] += sg * z.i; }
      else if constexpr ((k & 3) == 1) { ar[C] += sg * z.i; ai[C] -= sg * z.r; }
      else if constexpr ((k & 3) == 2) { ar[C] -= sg * z.r; ai[C] -= sg * z.i; }
      else { ar[C] -= sg * z.i; ai[C] += sg * z.r; }
    }
    rach_steer_acc<SPS, NCL, J, C + 1>(z, ar, ai);
  }
}
template <int SPS, int NCL, int J>
constexpr bool rach_steer_used() {
  constexpr int CJ = 64 / SPS;
  for (int c = 0; c < NCL; c++)
    if (J - CJ * c >= 0 && J - CJ * c <= 40) return true;
  return false;
}
// groups of TRX_RACH_STEER_GROUP j: the next group's reads are issued before this group's additions (double buffer), the
// accumulators are pinned after each group (hipcc otherwise sinks the accumulation chains below all 185 reads: 256 VGPRs)
#ifndef TRX_RACH_STEER_GROUP
#define TRX_RACH_STEER_GROUP 8
#endif
constexpr int kSteerG = TRX_RACH_STEER_GROUP;
template <int SPS, int NCL, int J0, int Q>
__device__ __forceinline__ void rach_steer_load1(const cx *Zl, cx (&z)[kSteerG]) {
  constexpr int JMAX = (64 / SPS) * (NCL - 1) + 40;
  if constexpr (Q < kSteerG) {
    z[Q] = mk(0, 0);
    if constexpr (J0 + Q <= JMAX) { if constexpr (rach_steer_used<SPS, NCL, J0 + Q>()) z[Q] = Zl[SPS * (J0 + Q)]; }
    rach_steer_load1<SPS, NCL, J0, Q + 1>(Zl, z);
  }
}
template <int SPS, int NCL, int J0>
__device__ __forceinline__ void rach_steer_load(const cx *Zl, cx (&z)[kSteerG]) { rach_steer_load1<SPS, NCL, J0, 0>(Zl, z); }
template <int SPS, int NCL, int J0, int Q>
__device__ __forceinline__ void rach_steer_group(const cx (&z)[kSteerG], float (&ar)[NCL], float (&ai)[NCL]) {
  if constexpr (Q < kSteerG) {
    rach_steer_acc<SPS, NCL, J0 + Q, 0>(z[Q], ar, ai);     // (k outside 0..40 for every c: nothing happens)
    rach_steer_group<SPS, NCL, J0, Q + 1>(z, ar, ai);
  }
}
template <int SPS, int NCL, int J0>
__device__ __forceinline__ void rach_steer(const cx *Zl, const cx (&zc)[kSteerG], float (&ar)[NCL], float (&ai)[NCL]) {
  constexpr int JMAX = (64 / SPS) * (NCL - 1) + 40;
  if constexpr (J0 <= JMAX) {
    cx zn[kSteerG];
    rach_steer_load<SPS, NCL, J0 + kSteerG>(Zl, zn);
    __builtin_amdgcn_sched_barrier(0);
    rach_steer_group<SPS, NCL, J0, 0>(zc, ar, ai);
#pragma unroll
    for (int c = 0; c < NCL; c++) { asm volatile("" : "+v"(ar[c])); asm volatile("" : "+v"(ai[c])); }
    __builtin_amdgcn_sched_barrier(0);
    rach_steer<SPS, NCL, J0 + kSteerG>(Zl, zn, ar, ai);
  }
}

// Where a burst's samples come from: packed complex float32 in memory, or (RachRxSrc) computed from the raw int16 stream
// of the receive front end, four multiply-adds per sample (trxsig_rxgen.h) -- the detector stages its burst twice and
// reads the energy window, that is all it ever asks of the samples.
//   stage<NIT, XF>(lane, xv): xv[it] = sample lane + 64 it - XF, zero outside [0, N)
struct RachMemSrc {
  const cx *x;
  int N;
  bool good;
  __device__ __forceinline__ RachMemSrc(const cx *samples, const int32_t *offset, const int32_t *length, int b, int sps) {
    const int off = offset[b];
    N = length[b];
    good = (off >= 0) && (N >= 92 * sps) && (N <= 157 * sps) && (N % sps == 0);
    x = samples + (good ? off : 0);
  }
  template <int NIT, int XF>
  __device__ __forceinline__ void stage(int lane, cx (&xv)[NIT]) const {
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int n = lane + 64 * it - XF;
      xv[it] = (n >= 0 && n < N) ? x[n] : mk(0, 0);
    }
  }
  __device__ __forceinline__ cx at(int n) const { return x[n]; }
};
struct RachRxSrc {
  RxGlobalSrc g;
  int N;
  bool good;
  __device__ __forceinline__ RachRxSrc(const TrxRxGen &gen, int b) : g(gen, b), N(g.u.N), good(true) {}
  template <int NIT, int XF>
  __device__ __forceinline__ void stage(int lane, cx (&xv)[NIT]) const {
    static_assert(XF < 3 * 64 - 63, "every lane is inside the burst from the third round on");
    RxIdx ix = g.index(0);
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int n = lane + 64 * it - XF;
      if (it < 3) ix = g.index(n >= 0 ? n : 0); else ix = rx_step<64>(ix);
      xv[it] = (n >= 0 && n < N) ? g.at(ix, n) : mk(0, 0);
    }
  }
  __device__ __forceinline__ cx at(int n) const { return g.at(g.index(n), n); }
};

#define RACH_SKIP (-1000)                                   // record marker: k_rach_front has already written this burst's outputs
// SPLIT: stop after step 2 and hand the exact neighbourhood, M, the energy and the three candidate valley sums
// (rint(toa) = M-1, M, M+1) to k_rach_peak2 through the record (rec: 25 complex slots, vsum: 3 float slots, [slot][Bpad]).
// NW = 2: a workgroup of two waves, a burst each, that SHARE the exact pass of step 2: a burst's contenders are its 26
// neighbourhood lags and (mostly) a handful more, so one wave's lanes 0..31 recompute this wave's lags and its lanes 32..63
// the other wave's (from the other burst's X) -- 1312 VALU instructions for two bursts instead of one.  A burst with more
// than six far contenders (or none at all) makes its workgroup fall back to a wave per burst.  `live` false (NW = 2 only):
// the odd wave of the last workgroup, which only tells its partner that it is not there.
template <int SPS, bool SPLIT, int NW, typename SRC>
__device__ __forceinline__ void rach_fast_burst(const int b, const bool live, const TrxTables *__restrict__ T, const SRC &src,
                                                float detect_thresh, float energy_thresh, float amp_err,
                                                uint8_t *__restrict__ flags, cx *__restrict__ amp_out,
                                                float *__restrict__ toa_out, float *__restrict__ avgpwr_out,
                                                cx *__restrict__ rec, float *__restrict__ vsum, int Bpad) {
  typedef RachGeom<SPS> R;
  typedef RachFast<SPS> Q;
  // LDS, 8.4 KB per wave (19 waves per CU; the kernel gains from every wave it can get: 366 us at 12 per CU, 415 at 9):
  //   A   the zero-padded burst X -- then, IN PLACE, its pulse-filtered copy Z for the steering pass -- then X again,
  //       re-staged from global memory (L2) for the tail corrections and the exact sums.  (X and Z side by side were 13 KB.)
  //   PWw the approximate powers the valley can read: lags M - 1 + 57 sps .. M + 1 + 107 sps, written once M is known;
  //   exv[64] exact correlation of the selected lags, exl[64] which lags they are, nb[26] exact neighbourhood
  //       corr[M-12..M+11] (+2 zero slots).
  constexpr int PWN = 50 * SPS + 8;                        // 107 sps - 57 sps + 1 terms, three candidate peaks
  static_assert(Q::ZPAD <= Q::XPAD, "Z is written over X");
  // (SPLIT: X is dead once the exact neighbourhood is known, and PWw is only written after that -- it takes X's place, and a
  //  paired workgroup needs 15.2 KB instead of 16.9: ten workgroups = 20 waves per CU instead of nine = 18)
  constexpr int PWS = SPLIT ? 0 : PWN;                     // floats of `side` that PWw takes
  constexpr int SIDE = PWS + 2 * 64 + 64;                  // (nb takes exl's place: the lag list is dead once M is known)
  __shared__ __attribute__((aligned(16))) cx xs[NW][Q::XPAD];
  __shared__ __attribute__((aligned(16))) float side_[NW][SIDE];
  __shared__ int pinfo[NW][4];                             // NW = 2: {can share the exact pass, lags listed, N}
  const int lane = threadIdx.x & 63;
  const int wave = NW == 1 ? 0 : (int)(threadIdx.x >> 6);
  float *const side = side_[wave];
  if constexpr (NW == 2) {
    if (!live) {
      if (lane == 0) pinfo[wave][0] = 0;
      wave_lds_fence();
      return;
    }
  }
#ifdef TRX_RACH_PROBE                                      // tools/rach_probe.py: clock64() stamps come back through avgpwr
  long long pt_[16] = {0};
  int pk_ = 0;
#define TRX_STAMP() pt_[pk_++] = clock64()
#define TRX_STAMP_FLUSH()                                                               \
  if (lane == 0 && avgpwr_out) {                                                        \
    long long v_ = 0;                                                                   \
    for (int k = 1; k < 8; k++) if ((b & 7) == k) v_ = pt_[k] - pt_[0];                 \
    avgpwr_out[b] = (float)v_;                                                          \
  }
#else
#define TRX_STAMP()
#define TRX_STAMP_FLUSH()
#endif
  TRX_STAMP();
  const int N = src.N;
  if (!src.good) {
    if (lane == 0) {
      flags[b] = TRXSIG_F_BADLEN; amp_out[b] = mk(0, 0); toa_out[b] = 0.0f; if (avgpwr_out) avgpwr_out[b] = 0.0f;
      if (SPLIT) rec[(size_t)24 * Bpad + b] = mk(__int_as_float(RACH_SKIP), 0.0f);
      if (NW == 2) pinfo[wave][0] = 0;
    }
    wave_lds_fence();
    return;
  }
  cx *X = xs[wave];
  cx *Z = xs[wave];                                        // the same storage, at different times
  float *PWw = SPLIT ? reinterpret_cast<float *>(xs[wave]) : side;
  cx *const exv_ = reinterpret_cast<cx *>(side + PWS);
  int *const exl_ = reinterpret_cast<int *>(side + PWS + 128);
  cx *const nb_ = reinterpret_cast<cx *>(side + PWS + 128);   // 26 complex over exl_'s 64 ints
  const cx *rseq = T->rach;

  float ex = 0.0f;
  {                                                        // every load in flight before the first LDS store
    constexpr int NIT = (Q::XPAD + 63) / 64;
    cx xv[NIT];
    src.template stage<NIT, Q::XF>(lane, xv);
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int i = lane + 64 * it;
      if (i < Q::XPAD) X[i] = xv[it];
    }
#pragma unroll
    for (int it = 0; it < NIT; it++) ex += norm2(xv[it]);  // burst energy (any order: it only scales an error bound)
  }
  ex = wave_sum_any_order(ex);                             // (DPP + readlane instead of six ds_bpermute round trips: trxsig_dev.h)
  const float dlt = amp_err * sqrtf(ex) * 1.001f;          // |approximate - reference| correlation amplitude, any lag
  float nrm[R::NEQ];
#pragma unroll
  for (int q = 0; q < R::NEQ; q++) {
    const int i = (lane & 15) + 16 * q;
    cx v = mk(0, 0);
    if (i < R::NE) v = src.at(i);
    nrm[q] = norm2(v);
  }
  float energy = energy_chain<SPS, 0>(0.0f, nrm);
  energy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(energy), 0));
  const bool energy_ok = energy_thresh < 0.0f || energy / (float)(unsigned)R::NE > energy_thresh * energy_thresh;
  if (!energy_ok) {                                        // Transceiver.cpp:298-306: correlator not run
    if (lane == 0) {
      flags[b] = 0; amp_out[b] = mk(0, 0); toa_out[b] = 0.0f; if (avgpwr_out) avgpwr_out[b] = energy / (float)(unsigned)R::NE;
      if (SPLIT) rec[(size_t)24 * Bpad + b] = mk(__int_as_float(RACH_SKIP), 0.0f);
      if (NW == 2) pinfo[wave][0] = 0;
    }
    wave_lds_fence();
    return;
  }
  wave_lds_fence();
  TRX_STAMP();                                             // 1: burst staged, energy done

  // ---- 1. approximate correlation at all lags (FMA allowed: this pass only steers) ----
  float pul[2 * SPS + 1];
#pragma unroll
  for (int j = 0; j < 2 * SPS + 1; j++) pul[j] = T->pulse[j];
  // two consecutive outputs per lane: Z[2g], Z[2g+1] from X[2g .. 2g+2sps+1] (16-byte LDS reads, each sample read once
  // per pair instead of once per output); per output the taps are still applied j ascending.
  // IN PLACE: an iteration's 64 lanes read positions 128 i .. 128 i + 135 and then write 128 i .. 128 i + 127; the LDS
  // executes a wave's instructions in order, so every lane has its samples before any lane's store lands, and the next
  // iteration only reads from 128 (i + 1) on -- which this one has not written.
  for (int g = lane; 2 * g < Q::ZPAD; g += 64) {
    cx xv[2 * SPS + 2];
    const float4 *xp4 = reinterpret_cast<const float4 *>(X + 2 * g);
#pragma unroll
    for (int q = 0; q < SPS + 1; q++) {
      const float4 t4 = xp4[q];
      xv[2 * q] = mk(t4.x, t4.y); xv[2 * q + 1] = mk(t4.z, t4.w);
    }
    float z0r = 0.0f, z0i = 0.0f, z1r = 0.0f, z1i = 0.0f;
#pragma unroll
    for (int j = 0; j < 2 * SPS + 1; j++) {
      z0r = fma_steer(pul[j], xv[j].r, z0r); z0i = fma_steer(pul[j], xv[j].i, z0i);
      z1r = fma_steer(pul[j], xv[j + 1].r, z1r); z1i = fma_steer(pul[j], xv[j + 1].i, z1i);
    }
    wave_lds_fence();                                      // (the loads above stay above the store)
    *reinterpret_cast<float4 *>(Z + 2 * g) = make_float4(z0r, z0i, z1r, z1i);
    wave_lds_fence();
  }
  wave_lds_fence();
  TRX_STAMP();                                             // 2: pulse filter done
  float pw[R::NCL];
  float bestP = 0.0f;
  int bestT = -1;
  {
    float ar[R::NCL], ai[R::NCL];
#pragma unroll
    for (int c = 0; c < R::NCL; c++) { ar[c] = 0.0f; ai[c] = 0.0f; }
    cx z0[kSteerG];
    rach_steer_load<SPS, R::NCL, 0>(Z + lane, z0);
    rach_steer<SPS, R::NCL, 0>(Z + lane, z0, ar, ai);
    // ---- the burst again (L2 by now) over its filtered copy: everything below works on X ----
    wave_lds_fence();                                      // everybody is done reading Z
    {
      constexpr int NIT = (Q::XPAD + 63) / 64;
      cx xv[NIT];
      src.template stage<NIT, Q::XF>(lane, xv);
#pragma unroll
      for (int it = 0; it < NIT; it++) {
        const int i = lane + 64 * it;
        if (i < Q::XPAD) X[i] = xv[it];
      }
    }
    wave_lds_fence();
    int opaque0 = 0;                                       // (single 8-byte LDS reads: see rach_exact_lag)
    asm volatile("" : "+s"(opaque0));
    const cx *Xo = X + opaque0;
#pragma unroll
    for (int c = 0; c < R::NCL; c++) {
      const int t = lane + 64 * c;
      float a_r = ar[c], a_i = ai[c];
      // pulse tails the reference's modulateBurst dropped: before symbol 0 (j < sps) and after symbol 40 (j = 2 sps)
      float e0r = 0.0f, e0i = 0.0f;
#pragma unroll
      for (int j = 0; j < SPS; j++) { const cx v = (j & 1) ? Xo[t + j] : X[t + j]; e0r = fma_steer(pul[j], v.r, e0r); e0i = fma_steer(pul[j], v.i, e0i); }
      const float s0 = (float)RachSym::v[0], s40 = (float)RachSym::v[40];
      a_r -= s0 * e0r; a_i -= s0 * e0i;                    // k = 0: conj(c_0) = s0
      const cx v40 = X[t + 42 * SPS];
      a_r -= s40 * pul[2 * SPS] * v40.r; a_i -= s40 * pul[2 * SPS] * v40.i;   // k = 40: i^40 = 1
      const float p = (t < N) ? a_r * a_r + a_i * a_i : -1.0f;
      pw[c] = p;
      if (p > bestP) { bestP = p; bestT = t; }
    }
  }
  wave_argmax(bestP, bestT);                               // larger power wins, equal powers: the lower lag

  TRX_STAMP();                                             // 3: approximate correlation + argmax done
  // ---- 2. exact recomputation of the contenders ----
  const int Ma = bestT;                                    // approximate argmax (-1: silence)
  float cut = bestP * (1.0f - RACH_DELTA);
  {
    const float ca = sqrtf(bestP) - 2.0f * dlt;            // a lag below this amplitude cannot hold the reference's maximum
    const float cb = ca > 0.0f ? ca * ca * (1.0f - 1e-6f) : -1.0f;   // (void: every lag contends -> the exact route below)
    cut = (cb < cut || !(dlt == dlt)) ? cb : cut;
  }
  const int nb0 = Ma - 13;                                 // neighbourhood lags nb0 .. nb0+25
  int nfar = 0;
  int *LG = exl_;
  if (lane < Q::NB) LG[lane] = nb0 + lane;
#pragma unroll
  for (int c = 0; c < R::NCL; c++) {
    const int t = lane + 64 * c;
    const bool far = (t < N) && (pw[c] >= cut) && (t < nb0 || t >= nb0 + Q::NB);
    const unsigned long long mask = __ballot(far);
    const int pos = nfar + __popcll(mask & ((1ull << lane) - 1ull));
    if (far && pos < 64 - Q::NB) LG[Q::NB + pos] = t;
    nfar += __popcll(mask);
  }
  wave_lds_fence();
  bool paired = false;
  if constexpr (NW == 2) {
    // (a wave that left early has said so before it ended, and the barrier does not wait for it)
    if (lane == 0) { pinfo[wave][0] = (Ma >= 0 && nfar <= 32 - Q::NB) ? 1 : 0; pinfo[wave][1] = Q::NB + nfar; pinfo[wave][2] = N; }
    __syncthreads();
    paired = pinfo[0][0] != 0 && pinfo[1][0] != 0;
  }
  int M;                                                   // exact argmax
  if (Ma < 0 || nfar > 64 - Q::NB) {
    // flat or silent burst: exact correlation at every lag (the k_rach_corr route)
    float bP = 0.0f; int bT = -1;
    for (int c = 0; c < R::NCL; c++) {
      const int t = lane + 64 * c;
      if (t < N) {
        const cx v = rach_exact_lag<SPS>(X, rseq, t);
        const float p = norm2(v);
        if (p > bP) { bP = p; bT = t; }
      }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
      const float oP = __shfl_xor(bP, m, 64);
      const int oT = __shfl_xor(bT, m, 64);
      const bool take = (oP > bP) || (oP == bP && oT >= 0 && (bT < 0 || oT < bT));
      if (take) { bP = oP; bT = oT; }
    }
    M = bT;
    if (lane < 24) {
      const int lag = M - 12 + lane;
      nb_[lane] = (lag >= 0 && lag < N) ? rach_exact_lag<SPS>(X, rseq, lag) : mk(0, 0);
    }
  } else {
    const int nl = Q::NB + nfar;
    const int t = lane < nl ? LG[lane] : -1;
    cx v = mk(0, 0);
    const bool valid = t >= 0 && t < N;
    if (NW == 2 && paired) {
      if (wave == 0) {                                     // both bursts' lags, 32 lanes each
        const int h = lane >> 5, l = lane & 31;
        const int *LGh = reinterpret_cast<const int *>(side_[h] + PWS + 128);
        const int th = l < pinfo[h][1] ? LGh[l] : -1;
        cx vh = mk(0, 0);
        if (th >= 0 && th < pinfo[h][2]) vh = rach_exact_lag<SPS>(xs[h], rseq, th);
        reinterpret_cast<cx *>(side_[h] + PWS)[l] = vh;
      }
      __syncthreads();
      if (valid) v = exv_[lane];                           // (valid: lane < nl <= 32)
    } else {
      if (valid) v = rach_exact_lag<SPS>(X, rseq, t);
      exv_[lane] = v;
    }
    float bP = valid ? norm2(v) : 0.0f;
    int bT = (valid && bP > 0.0f) ? t : -1;
    if (bT < 0) bP = 0.0f;
    wave_argmax(bP, bT);                                   // peakDetect's argmax: strict >, the first maximum wins (:672-678)
    M = bT;
    wave_lds_fence();
    if (M >= nb0 + 12 && M <= nb0 + 14) {                  // |M - Ma| <= 1: [M-12, M+11] lies inside the recomputed lags
      if (lane < 24) nb_[lane] = exv_[M - 12 + lane - nb0];
    } else if (lane < 24) {
      const int lag = M - 12 + lane;
      nb_[lane] = (lag >= 0 && lag < N) ? rach_exact_lag<SPS>(X, rseq, lag) : mk(0, 0);
    }
  }
  if (lane < 24) {                                         // interpolatePoint never uses the last sample (:646)
    const int lag = M - 12 + lane;
    if (lag > N - 2 || lag < 0) nb_[lane] = mk(0, 0);
  }
  if (lane >= 24 && lane < 26) nb_[lane] = mk(0, 0);
  // the approximate powers the valley can touch: lag M - 1 + 57 sps + w at PWw[w] (lags >= N hold -1 and are never read)
  const int pw0 = M - 1 + 57 * SPS;
#pragma unroll
  for (int c = 0; c < R::NCL; c++) {
    const int w = lane + 64 * c - pw0;
    if (w >= 0 && w < PWN) PWw[w] = pw[c];
  }
  wave_lds_fence();

  TRX_STAMP();                                             // 4: exact contenders + neighbourhood done
  if constexpr (SPLIT) {
    if (lane < 24) rec[(size_t)lane * Bpad + b] = nb_[lane];
    if (lane == 24) rec[(size_t)24 * Bpad + b] = mk(__int_as_float(M), energy);
    const int i0 = 57 * SPS, i1 = 107 * SPS;
    float vs[3];                                           // the approximate valley (step 4) for rint(toa) = M-1+a
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const int p = M - 1 + a;
      int last = N - 1 - p;
      if (last > i1) last = i1;
      vs[a] = 0.0f;
      for (int i = i0 + lane; i <= last; i += 64) vs[a] += PWw[p + i - pw0];
    }
#pragma unroll
    for (int a = 0; a < 3; a++) vs[a] = wave_sum_any_order(vs[a]);   // (approximate sums: rach_decide's bracket covers any order)
    if (lane < 4) vsum[(size_t)lane * Bpad + b] = lane == 0 ? vs[0] : (lane == 1 ? vs[1] : (lane == 2 ? vs[2] : dlt));
    TRX_STAMP();                                           // 5: record written
    TRX_STAMP_FLUSH();
    return;
  }
  // ---- 3. peakDetect's bisection on the exact neighbourhood (lanes 0..3) ----
  float peakIx, pkOwn, pkOther;
  quad_bisect<0, 1, false>(T, reinterpret_cast<const float *>(nb_), 0, lane & 3, M, 1 << 30, &peakIx, &pkOwn,
                           &pkOther);
  peakIx = __shfl(peakIx, 0, 64); pkOwn = __shfl(pkOwn, 0, 64); pkOther = __shfl(pkOther, 0, 64);
  const cx peak = mk(pkOwn, pkOther);

  TRX_STAMP();                                             // 5: bisection done
  // ---- 4. detectRACHBurst tail (:875-913) ----
  float toa = peakIx;
  cx amp = mk(0, 0);
  bool detected = false;
  if (!(toa < 0.0f) && !(toa > (float)N)) {
    const int p = (int)rintf(toa);
    const int i0 = 57 * SPS, i1 = 107 * SPS;
    int last = N - 1 - p;                                  // largest i with p + i < N
    if (last > i1) last = i1;
    const int cnt = last - i0 + 1;                         // numSamples
    if (cnt >= 2) {
      float vs = 0.0f;
      for (int i = i0 + lane; i <= last; i += 64) vs += PWw[p + i - pw0];
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) vs += __shfl_xor(vs, m, 64);
      const float peak_abs = sqrtf(norm2(peak));
      const int dec = rach_decide(peak_abs, vs, cnt, dlt, detect_thresh);
      detected = dec == 1;
      if (dec < 0) {
        // cannot be told from the approximate powers: the reference's valley, exactly (:888-901)
        float *VX = reinterpret_cast<float *>(exv_);
        float valley = 0.0f;
        for (int base = i0; base <= last; base += 64) {
          const int i = base + lane;
          float pv = 0.0f;
          if (i <= last) pv = norm2(rach_exact_lag<SPS>(X, rseq, p + i));
          wave_lds_fence();
          VX[lane] = pv;
          wave_lds_fence();
          if (lane == 0) {
            const int n = (last - base + 1) < 64 ? (last - base + 1) : 64;
            for (int k = 0; k < n; k++) valley += VX[k];
          }
        }
        valley = __shfl(valley, 0, 64);
        const float RMS = (float)((double)sqrtf(valley / (float)cnt) + 0.00001);
        detected = peak_abs / RMS > detect_thresh;
      }
      amp = cdiv(peak, T->rach_gain);                      // :905
      toa = toa - T->rach_toa - (float)(8 * SPS);          // :907
    }
  }
  if (lane == 0) {
    flags[b] = TRXSIG_F_ENERGY | (detected ? TRXSIG_F_DETECT : 0);
    amp_out[b] = amp;
    toa_out[b] = toa;
    if (avgpwr_out) avgpwr_out[b] = energy / (float)(unsigned)R::NE;
  }
  TRX_STAMP();                                             // 6: tail done
  TRX_STAMP_FLUSH();
#undef TRX_STAMP
#undef TRX_STAMP_FLUSH
}


// the whole detector in one launch, a wave per burst; with `list` (the hand-over of k_rach_peak2) the workgroups walk
// the *count listed bursts instead
template <int SPS>
__global__ __launch_bounds__(64) void k_rach_fast(const TrxTables *__restrict__ T, const cx *__restrict__ samples,
                                                   const int32_t *__restrict__ offset,
                                                   const int32_t *__restrict__ length, int B,
                                                   float detect_thresh, float energy_thresh, float amp_err,
                                                   uint8_t *__restrict__ flags, cx *__restrict__ amp_out,
                                                   float *__restrict__ toa_out, float *__restrict__ avgpwr_out,
                                                   const int32_t *__restrict__ list, const int32_t *__restrict__ count) {
  const int n = list ? (*count < B ? *count : B) : B;
  for (int i = blockIdx.x; i < n; i += gridDim.x) {        // (without a list: grid = B, one burst per workgroup)
    const int b = list ? list[i] : i;
    rach_fast_burst<SPS, false, 1>(b, true, T, RachMemSrc(samples, offset, length, b, SPS), detect_thresh, energy_thresh, amp_err, flags, amp_out,
                                toa_out, avgpwr_out, nullptr, nullptr, 0);
    wave_lds_fence();                                      // the next burst reuses the LDS
  }
}
// the same on bursts computed from the raw stream (hand-over list only)
template <int SPS>
__global__ __launch_bounds__(64) void k_rach_fast_rx(const TrxTables *__restrict__ T, TrxRxGen a, int B, float detect_thresh,
                                                      float energy_thresh, float amp_err, uint8_t *__restrict__ flags,
                                                      cx *__restrict__ amp_out, float *__restrict__ toa_out,
                                                      float *__restrict__ avgpwr_out, const int32_t *__restrict__ list,
                                                      const int32_t *__restrict__ count) {
  const int n = *count < B ? *count : B;
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const int b = list[i];
    rach_fast_burst<SPS, false, 1>(b, true, T, RachRxSrc(a, b), detect_thresh, energy_thresh, amp_err, flags, amp_out, toa_out, avgpwr_out,
                                nullptr, nullptr, 0);
    wave_lds_fence();
  }
}

#ifndef TRX_RACH_PAIR
#define TRX_RACH_PAIR 1   /* 1: two bursts per workgroup share the exact pass (rach_fast_burst, NW = 2); 0: a wave per burst (A/B) */
#endif
constexpr int kRachNW = TRX_RACH_PAIR ? 2 : 1;
// steps 1-2 of k_rach_fast (approximate correlation, exact contenders and neighbourhood) for every burst
template <int SPS>
// (81 VGPRs: five waves per SIMD, which is what ten 14.8 KB workgroups per CU need.  Forced to 80 for a sixth -- an eleventh
//  workgroup would fit the LDS -- the compiler spills 3 VGPRs and 11 SGPRs and the kernel takes 291 instead of 281 us.)
__global__ __launch_bounds__(64 * kRachNW) void k_rach_front(const TrxTables *__restrict__ T, const cx *__restrict__ samples,
                                                    const int32_t *__restrict__ offset,
                                                    const int32_t *__restrict__ length, int B, float energy_thresh, float amp_err,
                                                    uint8_t *__restrict__ flags, cx *__restrict__ amp_out,
                                                    float *__restrict__ toa_out, float *__restrict__ avgpwr_out,
                                                    cx *__restrict__ rec, float *__restrict__ vsum, int Bpad,
                                                    int32_t *__restrict__ count) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *count = 0;     // k_rach_peak2's hand-over list starts empty
  const int b = kRachNW * blockIdx.x + (threadIdx.x >> 6);   // two waves, a burst each (they share the exact pass)
  const bool live = b < B;
  rach_fast_burst<SPS, true, kRachNW>(b, live, T, RachMemSrc(samples, offset, length, live ? b : 0, SPS), 0.0f, energy_thresh, amp_err, flags,
                                amp_out, toa_out, avgpwr_out, rec, vsum, Bpad);
}
// ... on bursts computed from the raw int16 stream of the receive front end (no resampled stream in memory)
template <int SPS>
__global__ __launch_bounds__(64 * kRachNW) void k_rach_front_rx(const TrxTables *__restrict__ T, TrxRxGen a, int B, float energy_thresh,
                                                       float amp_err, uint8_t *__restrict__ flags, cx *__restrict__ amp_out,
                                                       float *__restrict__ toa_out, float *__restrict__ avgpwr_out,
                                                       cx *__restrict__ rec, float *__restrict__ vsum, int Bpad,
                                                       int32_t *__restrict__ count) {
  static_assert(SPS == 4, "the fused front end is the 260 : 96 resampler");
  if (blockIdx.x == 0 && threadIdx.x == 0) *count = 0;
  const int b = kRachNW * blockIdx.x + (threadIdx.x >> 6);
  const bool live = b < B;
  rach_fast_burst<SPS, true, kRachNW>(b, live, T, RachRxSrc(a, live ? b : 0), 0.0f, energy_thresh, amp_err, flags, amp_out, toa_out,
                                avgpwr_out, rec, vsum, Bpad);
}

// steps 3-4 with TWO lanes per burst (pair_bisect): peakDetect's bisection on the exact neighbourhood and
// detectRACHBurst's tail (:875-913) with the approximate valley.  A burst whose threshold lies inside the valley's error
// bar (rach_decide) goes on the hand-over list instead (k_rach_fast in list mode recomputes its valley exactly).
// (512 threads = 256 bursts round one LDS copy of the sinc table: see k_tsc_peak2)
constexpr int kRachPeak2Threads = 512;
template <int SPS>
__global__ __launch_bounds__(kRachPeak2Threads) void k_rach_peak2(const TrxTables *__restrict__ T, const cx *__restrict__ rec,
                                                    const float *__restrict__ vsum, const int32_t *__restrict__ length,
                                                    int Bpad, int B, float detect_thresh, uint8_t *__restrict__ flags,
                                                    cx *__restrict__ amp_out, float *__restrict__ toa_out,
                                                    float *__restrict__ avgpwr_out, int32_t *__restrict__ list,
                                                    int32_t *__restrict__ count) {
  typedef RachGeom<SPS> R;
  __shared__ __attribute__((aligned(16))) SincLds stab;
  const int tid = threadIdx.x;
  const int h = tid & 1;
  const int b = blockIdx.x * (kRachPeak2Threads / 2) + (tid >> 1);
  const bool live = b < B;
  const int bb = live ? b : B - 1;
  float4 tv[3072 / kRachPeak2Threads];
  sinc_lds_issue<kRachPeak2Threads>(T, tid, tv);
  const cx meta = rec[(size_t)24 * Bpad + bb];
  const int M = __float_as_int(meta.r);
  const float energy = meta.i;
  const int N = length[bb];
  cx q[23];                                                // pair_bisect's window: corr[M - 12 + k + 2h] (slots 24, 25: zeros)
#pragma unroll
  for (int k = 0; k < 23; k++) {
    const int ix = k + 2 * h;
    q[k] = (ix < 24) ? rec[(size_t)(ix < 24 ? ix : 23) * Bpad + bb] : mk(0, 0);
  }
  float vs3[3];
#pragma unroll
  for (int a = 0; a < 3; a++) vs3[a] = vsum[(size_t)a * Bpad + bb];
  const float dlt = vsum[(size_t)3 * Bpad + bb];
  sinc_lds_store<kRachPeak2Threads>(stab, tid, tv);
  __syncthreads();                                         // the only barrier

  int e;
  const cx peak = pair_bisect(stab, q, h, e);

  float toa = (float)(M - 1) + (float)e * 0.001953125f + 1.0f;       // peakIx = early + 1 (:699)
  cx amp = mk(0, 0);
  bool detected = false, handover = false;
  if (!(toa < 0.0f) && !(toa > (float)N)) {
    const int p = (int)rintf(toa);
    const int a = p - M + 1;                               // 0, 1 or 2
    const int i0 = 57 * SPS, i1 = 107 * SPS;
    int last = N - 1 - p;                                  // largest i with p + i < N
    if (last > i1) last = i1;
    const int cnt = last - i0 + 1;                         // numSamples
    if (cnt >= 2) {
      const float vs = a == 0 ? vs3[0] : (a == 1 ? vs3[1] : vs3[2]);
      const int dec = rach_decide(sqrtf(norm2(peak)), vs, cnt, dlt, detect_thresh);
      handover = dec < 0;
      detected = dec == 1;
      amp = cdiv(peak, T->rach_gain);                      // :905
      toa = toa - T->rach_toa - (float)(8 * SPS);          // :907
    }
  }
  if (live && h == 0 && M != RACH_SKIP) {
    if (handover) {
      list[atomicAdd(count, 1)] = b;                       // too close to call from approximate powers
    } else {
      flags[b] = TRXSIG_F_ENERGY | (detected ? TRXSIG_F_DETECT : 0);
      amp_out[b] = amp;
      toa_out[b] = toa;
#ifndef TRX_RACH_PROBE                                      // (the probe build keeps k_rach_front's stamps in avgpwr)
      if (avgpwr_out) avgpwr_out[b] = energy / (float)(unsigned)R::NE;
#endif
    }
  }
}


}  // namespace

// |db|_2 + 516 u |babs|_2 of the error model above, from the host copy of the tables, in double
float trx_rach_amp_err(const TrxTables *hT) {
  const int sps = (int)hT->sps, LB = 41 * sps;
  double db2 = 0.0, ba2 = 0.0;
  for (int m = 0; m < LB; m++) {
    double br = 0.0, bi = 0.0, ba = 0.0;                   // the implied sequence: sum over sps k + j - sps = m of sym_k i^k p[j]
    for (int k = 0; k <= 40; k++) {
      const int j = m + sps - sps * k;
      if (j < 0 || j > 2 * sps) continue;
      const double v = (double)RachSym::v[k] * (double)hT->pulse[j];
      switch (k & 3) { case 0: br += v; break; case 1: bi += v; break; case 2: br -= v; break; default: bi -= v; }
      ba += (double)hT->pulse[j] < 0 ? -(double)hT->pulse[j] : (double)hT->pulse[j];
    }
    const double dr = (double)hT->rach[m].r - br, di = (double)hT->rach[m].i - bi;
    db2 += dr * dr + di * di;
    ba2 += ba * ba;
  }
  const double u = 5.9604644775390625e-8;
  return (float)((sqrt(db2) + 516.0 * u * sqrt(ba2)) * 1.0001);
}

int trx_rach_rec_floats(int sps) {            // floats per burst in the rach record (complex slots + valley)
  switch (sps) {
    case 1: return 2 * RachGeom<1>::CSLOTS + RachGeom<1>::NVAL;
    case 2: return 2 * RachGeom<2>::CSLOTS + RachGeom<2>::NVAL;
    case 4: return 2 * RachGeom<4>::CSLOTS + RachGeom<4>::NVAL;
  }
  return 0;
}

#ifdef TRX_TUNING_BUILD
template <int S>
static void launch_rach_detect(hipStream_t st, const TrxTables *dT, const trx_c32 *samples, const int32_t *off,
                               const int32_t *len, int B, float detect_thresh, float energy_thresh, float *ws,
                               int Bpad, uint8_t *flags, trx_c32 *amp, float *toa, float *avgpwr,
                               TrxProfiler *prof) {
  trx_c32 *rec = (trx_c32 *)ws;
  float *recv = ws + (size_t)2 * RachGeom<S>::CSLOTS * Bpad;
  if (prof) prof->begin(TRXSIG_K_RACH_CORR, st);
  k_rach_corr<S><<<dim3((B + 3) / 4), dim3(256), 0, st>>>(dT, samples, off, len, B, rec, recv, Bpad);
  if (prof) { prof->end(TRXSIG_K_RACH_CORR, st); prof->begin(TRXSIG_K_RACH_PEAK, st); }
  k_rach_peak<S><<<dim3((B + 63) / 64), dim3(64), 0, st>>>(dT, rec, recv, len, Bpad, B, detect_thresh,
                                                           energy_thresh, flags, amp, toa, avgpwr);
  if (prof) prof->end(TRXSIG_K_RACH_PEAK, st);
}

#endif

template <int S>
static void launch_rach_fast(hipStream_t st, const TrxTables *dT, const trx_c32 *samples, const int32_t *off, const int32_t *len,
                             int B, float detect_thresh, float energy_thresh, float amp_err, float *ws, int Bpad, int split,
                             uint8_t *flags, trx_c32 *amp, float *toa, float *avgpwr, TrxProfiler *prof) {
  if (prof) prof->begin(TRXSIG_K_RACH_CORR, st);
  if (!split) {
    k_rach_fast<S><<<dim3(B), dim3(64), 0, st>>>(dT, samples, off, len, B, detect_thresh, energy_thresh, amp_err, flags, amp, toa,
                                                 avgpwr, nullptr, nullptr);
    if (prof) prof->end(TRXSIG_K_RACH_CORR, st);
    return;
  }
  // workspace: 25 complex + 4 float record slots (three candidate valley sums, the amplitude error bar), the hand-over
  // list and its counter (all [.][Bpad])
  trx_c32 *rec = (trx_c32 *)ws;
  float *vsum = ws + (size_t)2 * 25 * Bpad;
  int32_t *list = (int32_t *)(vsum + (size_t)4 * Bpad);
  int32_t *count = list + Bpad;
  k_rach_front<S><<<dim3((B + kRachNW - 1) / kRachNW), dim3(64 * kRachNW), 0, st>>>(dT, samples, off, len, B, energy_thresh, amp_err, flags, amp, toa, avgpwr, rec, vsum,
                                                Bpad, count);
  if (prof) { prof->end(TRXSIG_K_RACH_CORR, st); prof->begin(TRXSIG_K_RACH_PEAK, st); }
  k_rach_peak2<S><<<dim3((B + kRachPeak2Threads / 2 - 1) / (kRachPeak2Threads / 2)), dim3(kRachPeak2Threads), 0, st>>>(dT, rec, vsum, len, Bpad, B, detect_thresh, flags, amp, toa, avgpwr,
                                                               list, count);
  k_rach_fast<S><<<dim3(B < 512 ? B : 512), dim3(64), 0, st>>>(dT, samples, off, len, B, detect_thresh, energy_thresh, amp_err, flags,
                                                               amp, toa, avgpwr, list, count);
  if (prof) prof->end(TRXSIG_K_RACH_PEAK, st);
}

hipError_t trx_launch_rach_fast(hipStream_t st, int sps, const TrxTables *dT, const trx_c32 *samples,
                                const int32_t *off, const int32_t *len, int B, float detect_thresh,
                                float energy_thresh, float amp_err, float *ws, int Bpad, int split, uint8_t *flags, trx_c32 *amp,
                                float *toa, float *avgpwr, TrxProfiler *prof) {
  if (B <= 0) return hipSuccess;
  switch (sps) {
    case 1: launch_rach_fast<1>(st, dT, samples, off, len, B, detect_thresh, energy_thresh, amp_err, ws, Bpad, split, flags, amp, toa, avgpwr, prof); break;
    case 2: launch_rach_fast<2>(st, dT, samples, off, len, B, detect_thresh, energy_thresh, amp_err, ws, Bpad, split, flags, amp, toa, avgpwr, prof); break;
    case 4: launch_rach_fast<4>(st, dT, samples, off, len, B, detect_thresh, energy_thresh, amp_err, ws, Bpad, split, flags, amp, toa, avgpwr, prof); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// detectRACHBurst on bursts computed from the raw int16 stream (sps 4): k_rach_front_rx + k_rach_peak2 + hand-over; len[b] =
// the burst's length in samples (the slot schedule, as the front end cuts it)
hipError_t trx_launch_rx_rach(hipStream_t st, const TrxTables *dT, const TrxRxGen &gen, const int32_t *len, int B, float detect_thresh,
                              float energy_thresh, float amp_err, float *ws, int Bpad, uint8_t *flags, trx_c32 *amp, float *toa,
                              float *avgpwr, TrxProfiler *prof) {
  if (B <= 0) return hipSuccess;
  constexpr int S = 4;
  trx_c32 *rec = (trx_c32 *)ws;
  float *vsum = ws + (size_t)2 * 25 * Bpad;
  int32_t *list = (int32_t *)(vsum + (size_t)4 * Bpad);
  int32_t *count = list + Bpad;
  if (prof) prof->begin(TRXSIG_K_RACH_CORR, st);
  k_rach_front_rx<S><<<dim3((B + kRachNW - 1) / kRachNW), dim3(64 * kRachNW), 0, st>>>(dT, gen, B, energy_thresh, amp_err, flags, amp, toa, avgpwr, rec, vsum, Bpad, count);
  if (prof) { prof->end(TRXSIG_K_RACH_CORR, st); prof->begin(TRXSIG_K_RACH_PEAK, st); }
  k_rach_peak2<S><<<dim3((B + kRachPeak2Threads / 2 - 1) / (kRachPeak2Threads / 2)), dim3(kRachPeak2Threads), 0, st>>>(dT, rec, vsum, len, Bpad, B, detect_thresh, flags, amp, toa, avgpwr, list, count);
  k_rach_fast_rx<S><<<dim3(B < 512 ? B : 512), dim3(64), 0, st>>>(dT, gen, B, detect_thresh, energy_thresh, amp_err, flags, amp, toa, avgpwr,
                                                                  list, count);
  if (prof) prof->end(TRXSIG_K_RACH_PEAK, st);
  return hipGetLastError();
}

hipError_t trx_launch_rach_detect(hipStream_t st, int sps, const TrxTables *dT, const trx_c32 *samples,
                                  const int32_t *off, const int32_t *len, int B, float detect_thresh,
                                  float energy_thresh, float *ws, int Bpad, uint8_t *flags, trx_c32 *amp,
                                  float *toa, float *avgpwr, TrxProfiler *prof) {
#ifndef TRX_TUNING_BUILD
  (void)st; (void)sps; (void)dT; (void)samples; (void)off; (void)len; (void)B; (void)detect_thresh; (void)energy_thresh; (void)ws;
  (void)Bpad; (void)flags; (void)amp; (void)toa; (void)avgpwr; (void)prof;
  return hipErrorNotSupported;                             // the exact-at-every-lag route lives in the tuning build only
#else
  if (B <= 0) return hipSuccess;
  switch (sps) {
    case 1: launch_rach_detect<1>(st, dT, samples, off, len, B, detect_thresh, energy_thresh, ws, Bpad, flags, amp, toa, avgpwr, prof); break;
    case 2: launch_rach_detect<2>(st, dT, samples, off, len, B, detect_thresh, energy_thresh, ws, Bpad, flags, amp, toa, avgpwr, prof); break;
    case 4: launch_rach_detect<4>(st, dT, samples, off, len, B, detect_thresh, energy_thresh, ws, Bpad, flags, amp, toa, avgpwr, prof); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
#endif
}

