// trxsig_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the OpenBTS burst-processing path.
//
// Numerical contract: every float32 operation below is the reference's operation,
// in the reference's order, separately rounded.  This file is compiled with -ffp-contract=off
// (no v_fma/v_mac is ever formed from a*b+c) and with hipcc's default correctly-rounded
// division and square root, so the outputs are bit-identical to Transceiver/sigProcLib.cpp
// built for x86-64 (which has neither FMA contraction nor reassociation).  Where a sum's order
// is changed for parallelism the comment says why the result cannot change (only additions of
// +-0 are skipped or reordered).
//
// Work decomposition (64-wide wavefronts, no MFMA -- these are short O(N*K) filters):
//   k_tsc_corr   : 16 lanes (one DPP row) per burst, 4 bursts per wave.  Midamble correlation
//                  over the 36-symbol window with the 16 non-zero taps, energy detect, argmax,
//                  and a small per-burst record of the lags around the peak.
//   k_tsc_peak   : one LANE per burst.  The serial part of the reference (early-late bisection of
//                  peakDetect, valley RMS, threshold) has no parallelism inside a burst, so it is
//                  run for 64 bursts at once from the transposed records.
//   k_demod      : one wave per burst.  1/amp scaling, 21-tap fractional-delay filter evaluated
//                  only at the decimated symbol instants, reverse GMSK rotation, soft slicer.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include <type_traits>

#include "trxsig_tables.h"
#include "trxsig_launch.h"

#ifndef TRX_CORR_CG
#define TRX_CORR_CG 3
#endif
#ifndef TRX_CORR_WPS
#define TRX_CORR_WPS 1
#endif
#ifndef TRX_CORR_ROUNDS
#define TRX_CORR_ROUNDS 1
#endif
#ifndef TRX_DEMOD_WAVES
#define TRX_DEMOD_WAVES 4
#endif

namespace {

typedef trx_c32 cx;

__device__ __forceinline__ cx mk(float r, float i) { cx z; z.r = r; z.i = i; return z; }
// Complex<float>::operator* (Transceiver/Complex.h:83): (r*a.r - i*a.i, r*a.i + i*a.r)
__device__ __forceinline__ cx cmul(cx x, cx a) { return mk(x.r * a.r - x.i * a.i, x.r * a.i + x.i * a.r); }
__device__ __forceinline__ cx cmulr(cx x, float a) { return mk(x.r * a, x.i * a); }       // Complex.h:84
__device__ __forceinline__ cx cadd(cx x, cx a) { return mk(x.r + a.r, x.i + a.i); }
__device__ __forceinline__ float norm2(cx x) { return x.i * x.i + x.r * x.r; }            // Complex.h:119
__device__ __forceinline__ cx cinv(cx x) { float n = norm2(x); return mk(x.r / n, -x.i / n); }  // Complex.h:154-160
__device__ __forceinline__ cx cdiv(cx x, cx a) { return cmul(x, cinv(a)); }               // Complex.h:85

struct TapArg { float v[32]; };   // conj'd non-zero midamble taps passed as a kernel argument => SGPRs

// Tap classes.  The GMSK-rotated midamble taps are (+-1, eps) or (eps, +-1): one component is EXACTLY
// +-1 for most of them (the same ones for every training sequence -- it is a property of the rotation
// table), so the products with that component are exact and a*b + c with a single rounding (v_fma) is
// bit-identical to the reference's separately rounded multiply and add.  That saves 2 of the 8
// operations of a complex multiply-accumulate.  The class of every tap is a template parameter
// (2 bits per tap: 0 generic, 1 real part exact, 2 imaginary part exact); the host derives it from the
// actual taps and launches the generic instantiation whenever they do not match the expected pattern.
// These are the only FMAs outside division/sqrt expansions; tools/asm_stats.py recognises them by the
// marker comment.
#define TRX_TAPS_GENERIC 0u
template <int SPS> struct TapPattern {                     // taps 0,2,4.. real-exact, 1,3,5.. imaginary-exact
  static constexpr unsigned value = (SPS == 4) ? 0x19999999u : 0x99999999u;   // sps 4: tap 15 is (eps, -0.99999994)
};
__device__ __forceinline__ float fma_exact(float a, float b, float c) {       // a*b + c, a*b exact (b = +-1, SGPR)
  float r;
  asm("v_fma_f32 %0, %1, %2, %3 ; exact-product" : "=v"(r) : "v"(a), "s"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float fma_exact_subc(float a, float b, float c) {  // a*b - c
  float r;
  asm("v_fma_f32 %0, %1, %2, -%3 ; exact-product" : "=v"(r) : "v"(a), "s"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float fma_exact_negab(float a, float b, float c) { // c - a*b
  float r;
  asm("v_fma_f32 %0, %1, -%2, %3 ; exact-product" : "=v"(r) : "v"(a), "s"(b), "v"(c));
  return r;
}
// x * a as Complex<float>::operator* computes it (Complex.h:83), a in SGPRs, CLS = the tap's class
__device__ __forceinline__ cx cmul_tap(cx x, cx a, int cls) {
  if (cls == 1) {                                          // a.r = +-1: x.r*a.r and x.i*a.r are exact
    const float p = x.i * a.i, q = x.r * a.i;
    return mk(fma_exact_subc(x.r, a.r, p), fma_exact(x.i, a.r, q));
  }
  if (cls == 2) {                                          // a.i = +-1: x.i*a.i and x.r*a.i are exact
    const float p = x.r * a.r, q = x.i * a.r;
    return mk(fma_exact_negab(x.i, a.i, p), fma_exact(x.r, a.i, q));
  }
  return cmul(x, a);
}

#define TRX_PI_F 3.14159274101257324f             /* (float)M_PI, sigProcLib.cpp:43 */
#define TRX_2PI_F 6.28318548202514648f            /* (float)(2.0*M_PI), :44 */

// lane i of a 16-lane DPP row reads lane i+N of the same row (row_shl:N)
template <int N>
__device__ __forceinline__ float row_shl(float v) {
  if (N == 0) return v;
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + N, 0xf, 0xf, true));
}

__device__ __forceinline__ void wave_lds_fence() {
  // LDS operations of one wave execute in issue order; this only stops the compiler from
  // reordering them across the point where lanes start reading what other lanes wrote.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// sinLookup (sigProcLib.cpp:177-188) and sinc (:567-571) against the uploaded trig table
__device__ __forceinline__ float dev_sin_lookup(const float *__restrict__ sinT, float x) {
  float arg = x * (1 / TRX_2PI_F);
  while (arg > 1.0F) arg -= 1.0F;
  while (arg < 0.0F) arg += 1.0F;
  const float argT = arg * (float)TRX_TABLESIZE;
  const int argI = (int)argT;
  const float delta = argT - argI;
  const float iDelta = 1.0F - delta;
  return iDelta * sinT[argI] + delta * sinT[argI + 1];
}
__device__ __forceinline__ float dev_sinc(const float *__restrict__ sinT, float x) {
  if ((x >= 0.01F) || (x <= -0.01F)) return dev_sin_lookup(sinT, x) / x;
  return 1.0F;
}

// ---------------------------------------------------------------------------------------------
// k_tsc_corr: analyzeTrafficBurst's correlation (sigProcLib.cpp:951-955) + energyDetect (:916-932)
//   + peakDetect's argmax (:673-680).
//
// corr[t] = sum_j tmp[j]*w[t+s-j] with tmp = reverse(conj(midamble)), s = 8*sps-1, over the window
// w = burst[56*sps, 92*sps).  Only every sps-th midamble tap is non-zero, so with tap k = m/sps the
// sum is  sum_{k=15..0} w[t - 8*sps + sps*k] * conj(mid[sps*k])  in that order (ascending j).  The
// skipped taps are exact zeros: their products are +-0 and adding them never changes a value.
// Likewise out-of-range samples are read as 0 from the padded LDS window instead of being skipped.
//
// Lane r of a row handles lags t = r + 16*c.  For sps=4 the sample index r + 4*(4c+k) depends on
// (c,k) only through 4c+k, so the 144 (c,k) pairs touch 48 distinct LDS words per lane.
// ---------------------------------------------------------------------------------------------
template <int SPS>
struct CorrGeom {
  static constexpr int NL = 36 * SPS;                    // lags (window length)
  static constexpr int NC = (NL + 15) / 16;              // lags per lane
  static constexpr int FRONT = 8 * SPS;                  // zero pad in front of the window
  static constexpr int WPAD0 = NC * 16 + 15 * SPS + 1;   // padded window length needed
  static constexpr int WPAD = ((WPAD0 + 15) / 32) * 32 + 16;   // rounded up to 16 (mod 32): rows 32 dwords apart (mod 64)
  static constexpr int H = (5 * SPS + 1 > 12) ? 5 * SPS + 1 : 12;   // record half width
  static constexpr int NS = 2 * H + 1;                   // corr slots in a record (+1 meta slot)
  static constexpr int NE = 20 * SPS;                    // energyDetect window
  static constexpr int NEQ = (NE + 15) / 16;
};

// energy += norm2(x[I]) for I = 0 .. NE-1 strictly in order; norm I lives in lane I%16 of
// register nrm[I/16], and lane 0 of the row pulls it over with a DPP row shift.
// One instruction per step: v_add_f32 with the DPP row shift on the incoming operand (hipcc does not
// fold v_mov_dpp into the add and would hoist all 80 moves, costing 80 VGPRs).  The DPP operand
// (nrm) is written long before the chain; the s_nop covers the VALU-write -> DPP-read wait states
// that hipcc does not insert around inline asm.
template <int N>
__device__ __forceinline__ float add_row_shl(float acc, float v) {
  float r;
  if (N == 0) asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(acc));
  else asm volatile("v_add_f32_dpp %0, %1, %2 row_shl:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                    : "=v"(r) : "v"(v), "v"(acc), "n"(N));
  return r;
}
template <int SPS, int I>
__device__ __forceinline__ float energy_chain_step(float acc, const float (&nrm)[CorrGeom<SPS>::NEQ]) {
  if constexpr (I < CorrGeom<SPS>::NE) {
    acc = add_row_shl<I % 16>(acc, nrm[I / 16]);
    return energy_chain_step<SPS, I + 1>(acc, nrm);
  } else {
    return acc;
  }
}
template <int SPS, int I>
__device__ __forceinline__ float energy_chain(float acc, const float (&nrm)[CorrGeom<SPS>::NEQ]) {
  asm volatile("s_nop 1");
  return energy_chain_step<SPS, I>(acc, nrm);
}

// One round = 4 bursts of one wave.  corr_issue puts a round's global loads in flight (8 bytes per lane
// per load, any alignment); corr_round consumes them.  k_tsc_corr issues the loads of BOTH of its
// rounds before working on the first, so the second round's HBM latency hides under arithmetic
// (workgroups of a launch otherwise march through their load and math phases in lockstep: the phase
// costs of the single-round kernel measured perfectly additive).
template <int SPS>
struct CorrIn {
  static constexpr int NW = (CorrGeom<SPS>::NL + 15) / 16;     // window samples per lane
  cx w[NW];
  cx e[CorrGeom<SPS>::NEQ];
  int b;
  bool live, good;
};

template <int SPS>
__device__ __forceinline__ void corr_issue(CorrIn<SPS> &in, int b, int B, int r, const cx *__restrict__ samples,
                                           const int32_t *__restrict__ offset, const int32_t *__restrict__ length) {
  typedef CorrGeom<SPS> G;
  in.b = b;
  in.live = b < B;
  int off = 0, len = 0;
  if (in.live) { off = offset[b]; len = length[b]; }
  in.good = in.live && (off >= 0) && (len >= 92 * SPS) && (len <= 157 * SPS) && (len % SPS == 0);
  const cx *x = samples + (in.good ? off : 0);
#pragma unroll
  for (int i = 0; i < CorrIn<SPS>::NW; i++) {
    const int q = r + 16 * i;
    in.w[i] = (in.good && q < G::NL) ? x[56 * SPS + q] : mk(0, 0);
  }
#pragma unroll
  for (int q = 0; q < G::NEQ; q++) {
    const int i = r + 16 * q;
    in.e[q] = (in.good && i < G::NE) ? x[i] : mk(0, 0);
  }
}

// REC: write the detect->peak record (k_tsc_corr); otherwise the correlation just stays in W[0, NL)
// (k_normal_quad).  M_out / energy_out: argmax lag and energy sum of the lane's burst.
// EFIRST: E aliases the row (k_normal_quad): the energy window's norms are staged, summed and done
// with before the correlation window is written over them.
template <int SPS, bool REC, bool EFIRST = false, unsigned TAPCLS = TRX_TAPS_GENERIC>
__device__ __forceinline__ void corr_round(const CorrIn<SPS> &in, cx *W, float4 *E, int lane, int r, const cx (&tap)[16],
                                           cx *__restrict__ rec, int Bpad, int &M_out, float &energy_out) {
  typedef CorrGeom<SPS> G;
  auto stage_norms = [&] {
    float *ef = reinterpret_cast<float *>(E);
#pragma unroll
    for (int q = 0; q < G::NEQ; q++) {
      const int i = r + 16 * q;
      if (i < G::NE) ef[i] = norm2(in.e[q]);
    }
  };
  // energyDetect: energy += norm2(x[i]), i = 0 .. 20*sps-1, strictly in order (:925-928).  Every
  // lane of the row adds the norms up sequentially (same address in a row -> broadcast reads);
  // a DPP row-shift chain does the same but issues ~5x slower per step.
  auto sum_norms = [&] {
    float energy = 0.0f;
#pragma unroll
    for (int i4 = 0; i4 < (G::NE + 3) / 4; i4++) {
      const float4 e = E[i4];
      energy = energy + e.x;
      if (4 * i4 + 1 < G::NE) energy = energy + e.y;
      if (4 * i4 + 2 < G::NE) energy = energy + e.z;
      if (4 * i4 + 3 < G::NE) energy = energy + e.w;
    }
    return energy;
  };
  float energy = 0.0f;
  if (EFIRST) {
    stage_norms();
    wave_lds_fence();
    energy = sum_norms();
    asm volatile("" : "+v"(energy));                       // the sum is complete here, before the norms are overwritten
    wave_lds_fence();
  }
  // ---- window (zero padded) and the energy window's norms into LDS ----
  for (int q = r; q < G::FRONT; q += 16) W[q] = mk(0, 0);
  for (int q = G::FRONT + G::NL + r; q < G::WPAD; q += 16) W[q] = mk(0, 0);
#pragma unroll
  for (int i = 0; i < CorrIn<SPS>::NW; i++) {
    const int q = r + 16 * i;
    if (q < G::NL) W[G::FRONT + q] = in.w[i];
  }
  if (!EFIRST) stage_norms();
  wave_lds_fence();
  if (!EFIRST) energy = sum_norms();

  // ---- correlation: 16 non-zero taps, k descending = j ascending ----
  float bestP = 0.0f;
  int bestT = -1;
  cx cval[G::NC];
  constexpr int CG = (SPS == 4) ? TRX_CORR_CG : 1;         // lags per register group
  constexpr int UPC = 16 / SPS;                            // stride-SPS sample steps per 16 lags
  constexpr int NU = UPC * (CG - 1) + 16;
#pragma unroll
  for (int c0 = 0; c0 < G::NC; c0 += CG) {
    cx sv[NU];                                             // sv[u] = W[r + 16*c0 + SPS*u]
#pragma unroll
    for (int u = 0; u < NU; u++) sv[u] = W[r + 16 * c0 + SPS * u];
#pragma unroll
    for (int cc = 0; cc < CG; cc++) {
      if (c0 + cc < G::NC) {
        cx acc = mk(0, 0);
#pragma unroll
        for (int k = 15; k >= 0; k--) acc = cadd(acc, cmul_tap(sv[UPC * cc + k], tap[k], (TAPCLS >> (2 * k)) & 3));
        cval[c0 + cc] = acc;
      }
    }
  }
  wave_lds_fence();                                        // every lane is done reading the window
#pragma unroll
  for (int c = 0; c < G::NC; c++) {
    const int t = r + 16 * c;
    if (t < G::NL) {
      W[t] = cval[c];
      const float p = norm2(cval[c]);
      if (p > bestP) { bestP = p; bestT = t; }             // strict >, first maximum (:675)
    }
  }
  // first maximum over the row: larger power wins, equal power -> smaller lag
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) {
    const float oP = __shfl_xor(bestP, m, 64);
    const int oT = __shfl_xor(bestT, m, 64);
    const bool take = (oP > bestP) || (oP == bestP && oT >= 0 && (bestT < 0 || oT < bestT));
    if (take) { bestP = oP; bestT = oT; }
  }
  wave_lds_fence();

  M_out = bestT;
  energy_out = energy;
  // ---- record: corr[M-H .. M+H] (zeros outside [0,NL)), then {M, energy} ----
  if (REC && in.live) {
    const int M = bestT;
    for (int s = r; s <= G::NS; s += 16) {
      cx v = mk(0, 0);
      if (s < G::NS) {
        const int lag = M - G::H + s;
        if (lag >= 0 && lag < G::NL) v = W[lag];
      } else {
        v = mk(__int_as_float(in.good ? M : -2), energy);  // M = -2 marks an invalid burst
      }
      rec[(size_t)s * Bpad + in.b] = v;
    }
  }
  wave_lds_fence();                                        // record reads done before the row is reused
  (void)lane;
}

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

  CorrIn<SPS> in[TRX_CORR_ROUNDS];
#pragma unroll
  for (int i = 0; i < TRX_CORR_ROUNDS; i++)
    corr_issue<SPS>(in[i], (blockIdx.x * TRX_CORR_ROUNDS + i) * 16 + slot, B, r, samples, offset, length);
#pragma unroll
  for (int i = 0; i < TRX_CORR_ROUNDS; i++) {
    int M;
    float energy;
    corr_round<SPS, true, true, TAPCLS>(in[i], rows[slot], reinterpret_cast<float4 *>(rows[slot]), lane, r, tap, rec, Bpad, M, energy);
  }
}

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
  auto load_row = [&](int f, float (&s)[24]) {
    const float4 *row = reinterpret_cast<const float4 *>(tab[f & 511]);
#pragma unroll
    for (int q = 0; q < 6; q++) {
      const float4 v = row[q];
      s[4 * q] = v.x; s[4 * q + 1] = v.y; s[4 * q + 2] = v.z; s[4 * q + 3] = v.w;
    }
  };
  // pa = interpolatePoint(ix), pb = interpolatePoint(ix + dI2): same fractional part, row s
  auto interp2 = [&](float ix, int dI2, const float (&s)[24], cx &pa, cx &pb) {
    const int I = (int)floorf(ix);
    int base = I - M + 2;                                  // loc index of tap j = 0 (0..3 by construction)
    base = base < 0 ? 0 : (base > 3 ? 3 : base);
    pa = mk(0, 0); pb = mk(0, 0);
#pragma unroll
    for (int j = 0; j < 21; j++) {
      pa = cadd(pa, cmulr(loc[base + j][lane], s[j]));
      pb = cadd(pb, cmulr(loc[base + j + dI2][lane], s[j]));
    }
  };
  auto frac512 = [](float ix) { return (int)((ix - floorf(ix)) * 512.0f); };

  float early = (float)M - 1;
  float incr = 0.5f;
  bool active = true;
  float cur[24], up[24], dn[24];
  load_row(0, cur);                                        // early = M-1 is an integer
#pragma unroll 1
  for (int step = 0; step < 9; step++) {                   // incr = 2^-1 .. 2^-9  (> 1/1024)
    load_row(frac512(early + incr), up);                   // candidates for the next step / the final point
    load_row(frac512(early - incr), dn);
    cx e, l;
    interp2(early, 2, cur, e, l);
    const float ne = norm2(e), nl = norm2(l);
    const bool goUp = ne < nl, goDn = ne > nl;
    if (active) {
      if (goUp) early += incr;
      else if (goDn) early -= incr;
      else active = false;                                 // "else break" (:695)
      if (active) incr = incr * 0.5f;
    }
    const bool moved = active;                             // row changes only if the index moved
#pragma unroll
    for (int j = 0; j < 24; j++) cur[j] = moved ? (goUp ? up[j] : dn[j]) : cur[j];
  }
  *peakIx = early + 1.0f;                                  // same fractional part as `early`: row = cur
  cx peak, dummy;
  interp2(*peakIx, 0, cur, peak, dummy);
  return peak;
}

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
  const cx meta = rec[(size_t)G::NS * Bpad + bb];
  const int M = __float_as_int(meta.r);
  const float energy = meta.i;
  const bool good = M != -2;
#pragma unroll
  for (int s = 0; s < G::NS; s++) {
    const cx v = rec[(size_t)s * Bpad + bb];
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
//  4. the valley RMS uses the approximate powers; if peak/RMS lands within 1e-3 (relative) of the
//     threshold -- where a 1e-5 error could matter -- the valley lags are recomputed exactly and summed
//     in the reference's order, so the detect decision is the reference's in every case.
// If more far-away candidates turn up than fit in one pass (flat noise, silence) the burst takes
// the exact route for all lags.
// ---------------------------------------------------------------------------------------------
#define RACH_DELTA 4e-3f
#define RACH_GUARD 1e-3f
__device__ __constant__ const signed char kRachSym[41] = {           // 2*bit-1 of gRACHSynchSequence (GSM/GSMCommon.cpp:57)
  -1, 1, -1, -1, 1, -1, 1, 1, -1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, -1, 1, -1, -1, -1, 1, 1, 1, 1, -1,
  -1, -1 };

template <int SPS>
struct RachFast {
  typedef RachGeom<SPS> R;
  static constexpr int XF = R::F + SPS;                     // X[i] = x[i - XF]
  static constexpr int XPAD = 64 * R::NCL + R::LB + 4 * SPS + 8;
  static constexpr int ZPAD = 64 * R::NCL + 40 * SPS + 1;   // Zs[i] = sum_j p[j] X[i+j]
  static constexpr int NB = 26;                             // lags M~-13 .. M~+12 always recomputed
};

// exact corr[t] (sigProcLib.cpp:474-503 + 322-366): sum over m = LB-1 .. 0 of x[t-F+m]*conj(rach[m])
template <int SPS>
__device__ __forceinline__ cx rach_exact_lag(const cx *X, const cx *__restrict__ rseq, int t) {
  typedef RachGeom<SPS> R;
  cx acc = mk(0, 0);
  const cx *xp = X + t + SPS;                              // X index of x[t-F+m] is t + m + SPS
#pragma unroll 4
  for (int m = R::LB - 1; m >= 0; m--) {
    const cx rm = rseq[m];
    acc = cadd(acc, cmul(xp[m], mk(rm.r, -rm.i)));
  }
  return acc;
}

template <int SPS>
__global__ __launch_bounds__(64) void k_rach_fast(const TrxTables *__restrict__ T, const cx *__restrict__ samples,
                                                   const int32_t *__restrict__ offset,
                                                   const int32_t *__restrict__ length, int B,
                                                   float detect_thresh, float energy_thresh,
                                                   uint8_t *__restrict__ flags, cx *__restrict__ amp_out,
                                                   float *__restrict__ toa_out, float *__restrict__ avgpwr_out) {
  typedef RachGeom<SPS> R;
  typedef RachFast<SPS> Q;
  __shared__ cx xs[1][Q::XPAD];
  __shared__ cx zs[1][Q::ZPAD];                             // pulse-filtered burst; later approx powers (float view)
  __shared__ cx exv[1][64];                                 // exact correlation of the selected lags
  __shared__ int exl[1][64];                                // ... and which lags they are
  __shared__ cx nb[1][26];                                  // exact neighbourhood corr[M-12..M+11] (+2 zero slots)

  const int lane = threadIdx.x;
  constexpr int wave = 0;                                  // one wave per workgroup (14 KB of LDS each)
  const int b = blockIdx.x;
  if (b >= B) return;
  const int off = offset[b], N = length[b];
  const bool good = (off >= 0) && (N >= 92 * SPS) && (N <= 157 * SPS) && (N % SPS == 0);
  if (!good) {
    if (lane == 0) { flags[b] = TRXSIG_F_BADLEN; amp_out[b] = mk(0, 0); toa_out[b] = 0.0f; if (avgpwr_out) avgpwr_out[b] = 0.0f; }
    return;
  }
  const cx *x = samples + off;
  cx *X = xs[wave];
  cx *Z = zs[wave];
  float *PW = reinterpret_cast<float *>(Z);
  const cx *rseq = T->rach;

  for (int i = lane; i < Q::XPAD; i += 64) {
    const int n = i - Q::XF;
    X[i] = (n >= 0 && n < N) ? x[n] : mk(0, 0);
  }
  float nrm[R::NEQ];
#pragma unroll
  for (int q = 0; q < R::NEQ; q++) {
    const int i = (lane & 15) + 16 * q;
    cx v = mk(0, 0);
    if (i < R::NE) v = x[i];
    nrm[q] = norm2(v);
  }
  float energy = energy_chain<SPS, 0>(0.0f, nrm);
  energy = __shfl(energy, 0, 64);
  const bool energy_ok = energy_thresh < 0.0f || energy / (float)(unsigned)R::NE > energy_thresh * energy_thresh;
  if (!energy_ok) {                                        // Transceiver.cpp:298-306: correlator not run
    if (lane == 0) { flags[b] = 0; amp_out[b] = mk(0, 0); toa_out[b] = 0.0f; if (avgpwr_out) avgpwr_out[b] = energy / (float)(unsigned)R::NE; }
    return;
  }
  wave_lds_fence();

  // ---- 1. approximate correlation at all lags (FMA allowed: this pass only steers) ----
  float pul[2 * SPS + 1];
#pragma unroll
  for (int j = 0; j < 2 * SPS + 1; j++) pul[j] = T->pulse[j];
  for (int i = lane; i < Q::ZPAD; i += 64) {
    float zr = 0.0f, zi = 0.0f;
#pragma unroll
    for (int j = 0; j < 2 * SPS + 1; j++) {
      const cx v = X[i + j];
      zr = __builtin_fmaf(pul[j], v.r, zr); zi = __builtin_fmaf(pul[j], v.i, zi);
    }
    Z[i] = mk(zr, zi);
  }
  wave_lds_fence();
  float pw[R::NCL];
  float bestP = 0.0f;
  int bestT = -1;
#pragma unroll
  for (int c = 0; c < R::NCL; c++) {
    const int t = lane + 64 * c;
    float ar = 0.0f, ai = 0.0f;
#pragma unroll
    for (int k = 0; k < 41; k++) {                         // conj(c_k) z, c_k = sym_k i^k: additions only
      const cx z = Z[t + SPS * k];
      const float sg = (float)kRachSym[k];
      if ((k & 3) == 0) { ar += sg * z.r; ai += sg * z.i; }
      else if ((k & 3) == 1) { ar += sg * z.i; ai -= sg * z.r; }
      else if ((k & 3) == 2) { ar -= sg * z.r; ai -= sg * z.i; }
      else { ar -= sg * z.i; ai += sg * z.r; }
    }
    // pulse tails the reference's modulateBurst dropped: before symbol 0 (j < sps) and after symbol 40 (j = 2 sps)
    float e0r = 0.0f, e0i = 0.0f;
#pragma unroll
    for (int j = 0; j < SPS; j++) { const cx v = X[t + j]; e0r = __builtin_fmaf(pul[j], v.r, e0r); e0i = __builtin_fmaf(pul[j], v.i, e0i); }
    const float s0 = (float)kRachSym[0], s40 = (float)kRachSym[40];
    ar -= s0 * e0r; ai -= s0 * e0i;                        // k = 0: conj(c_0) = s0
    const cx v40 = X[t + 42 * SPS];
    ar -= s40 * pul[2 * SPS] * v40.r; ai -= s40 * pul[2 * SPS] * v40.i;   // k = 40: i^40 = 1
    const float p = (t < N) ? ar * ar + ai * ai : -1.0f;
    pw[c] = p;
    if (p > bestP) { bestP = p; bestT = t; }
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) {
    const float oP = __shfl_xor(bestP, m, 64);
    const int oT = __shfl_xor(bestT, m, 64);
    const bool take = (oP > bestP) || (oP == bestP && oT >= 0 && (bestT < 0 || oT < bestT));
    if (take) { bestP = oP; bestT = oT; }
  }
  wave_lds_fence();                                        // everybody is done reading Z
#pragma unroll
  for (int c = 0; c < R::NCL; c++) PW[lane + 64 * c] = pw[c];   // approximate powers (lags >= N hold -1)

  // ---- 2. exact recomputation of the contenders ----
  const int Ma = bestT;                                    // approximate argmax (-1: silence)
  const float cut = bestP * (1.0f - RACH_DELTA);
  const int nb0 = Ma - 13;                                 // neighbourhood lags nb0 .. nb0+25
  int nfar = 0;
  int *LG = exl[wave];
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
      nb[wave][lane] = (lag >= 0 && lag < N) ? rach_exact_lag<SPS>(X, rseq, lag) : mk(0, 0);
    }
  } else {
    const int nl = Q::NB + nfar;
    const int t = lane < nl ? LG[lane] : -1;
    cx v = mk(0, 0);
    const bool valid = t >= 0 && t < N;
    if (valid) v = rach_exact_lag<SPS>(X, rseq, t);
    exv[wave][lane] = v;
    float bP = valid ? norm2(v) : 0.0f;
    int bT = (valid && bP > 0.0f) ? t : -1;
    if (bT < 0) bP = 0.0f;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
      const float oP = __shfl_xor(bP, m, 64);
      const int oT = __shfl_xor(bT, m, 64);
      const bool take = (oP > bP) || (oP == bP && oT >= 0 && (bT < 0 || oT < bT));
      if (take) { bP = oP; bT = oT; }
    }
    M = bT;
    wave_lds_fence();
    if (M >= nb0 + 12 && M <= nb0 + 14) {                  // |M - Ma| <= 1: [M-12, M+11] lies inside the recomputed lags
      if (lane < 24) nb[wave][lane] = exv[wave][M - 12 + lane - nb0];
    } else if (lane < 24) {
      const int lag = M - 12 + lane;
      nb[wave][lane] = (lag >= 0 && lag < N) ? rach_exact_lag<SPS>(X, rseq, lag) : mk(0, 0);
    }
  }
  if (lane < 24) {                                         // interpolatePoint never uses the last sample (:646)
    const int lag = M - 12 + lane;
    if (lag > N - 2 || lag < 0) nb[wave][lane] = mk(0, 0);
  }
  if (lane >= 24 && lane < 26) nb[wave][lane] = mk(0, 0);
  wave_lds_fence();

  // ---- 3. peakDetect's bisection on the exact neighbourhood (lanes 0..3) ----
  float peakIx, pkOwn, pkOther;
  quad_bisect<0, 1, false>(T, reinterpret_cast<const float *>(nb[wave]), 0, lane & 3, M, 1 << 30, &peakIx, &pkOwn,
                           &pkOther);
  peakIx = __shfl(peakIx, 0, 64); pkOwn = __shfl(pkOwn, 0, 64); pkOther = __shfl(pkOther, 0, 64);
  const cx peak = mk(pkOwn, pkOther);

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
      for (int i = i0 + lane; i <= last; i += 64) vs += PW[p + i];
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) vs += __shfl_xor(vs, m, 64);
      float RMS = (float)((double)sqrtf(vs / (float)cnt) + 0.00001);
      float peakToMean = sqrtf(norm2(peak)) / RMS;
      if (fabsf(peakToMean - detect_thresh) <= RACH_GUARD * fabsf(detect_thresh) || !(vs == vs)) {
        // too close to call from approximate powers: the reference's valley, exactly (:888-901)
        float *VX = reinterpret_cast<float *>(exv[wave]);
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
        RMS = (float)((double)sqrtf(valley / (float)cnt) + 0.00001);
        peakToMean = sqrtf(norm2(peak)) / RMS;
      }
      amp = cdiv(peak, T->rach_gain);                      // :905
      toa = toa - T->rach_toa - (float)(8 * SPS);          // :907
      detected = peakToMean > detect_thresh;
    }
  }
  if (lane == 0) {
    flags[b] = TRXSIG_F_ENERGY | (detected ? TRXSIG_F_DETECT : 0);
    amp_out[b] = amp;
    toa_out[b] = toa;
    if (avgpwr_out) avgpwr_out[b] = energy / (float)(unsigned)R::NE;
  }
}

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
template <int SPS, bool RAW, int NSMAX>
__device__ __forceinline__ void demod_core(const TrxTables *__restrict__ T, cx *P, const cx *xb, int N, bool wide,
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
    if ((float)f == f512) {                                // on the 1/512 grid: sinc_grid[f][j] (uniform -> s_load)
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
      if (u0 >= 0 && u0 < G::U) P[(u0 % SPS) * G::QLEN + u0 / SPS] = cmul(xb[N - 1], inv);
    }
  } else {
    for (int n = lane; n < N; n += 64) {
      const int u0 = n + lo;
      if (u0 >= 0 && u0 < G::U) P[(u0 % SPS) * G::QLEN + u0 / SPS] = cmul(xb[n], inv);
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

template <int SPS, int LPB> struct FusedGeom;
template <int SPS, int LPB, typename HOOK>
__device__ __forceinline__ void fused_demod(const TrxTables *__restrict__ T, cx *P, const float4 (&v)[(157 * SPS / 2 + LPB - 1) / LPB],
                                            int N, cx amp, float toa, int hl, float *sb, uint8_t *hbp, int nsoft, HOOK staged,
                                            const float *tp_pre, const cx *rv_pre);

template <int SPS, bool RAW, int NSMAX>
__global__ __launch_bounds__(64 * TRX_DEMOD_WAVES) void k_demod(const TrxTables *__restrict__ T,
                                               const cx *__restrict__ samples,
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
  const cx *xb = samples + off;

  // ---- issue the burst's loads first (16-byte path: 2 samples per lane per load) ----
  constexpr int NLD = (157 * SPS / 2 + 63) / 64;           // float4 loads per lane
  const bool wide = (off & 1) == 0;
  float4 v[NLD];
  if (wide) {
    const float4 *xv = reinterpret_cast<const float4 *>(xb);
#pragma unroll
    for (int i = 0; i < NLD; i++) {
      const int q = lane + 64 * i;
      v[i] = (q < N / 2) ? xv[q] : make_float4(0, 0, 0, 0);
    }
  }
  // the common case (148 soft bits, even offset and length) goes through fused_demod: same arithmetic,
  // but a lane owns three CONSECUTIVE soft bits, whose filter windows share 34 of their 63 staged words
  // (measured: 68.0 -> 64.6 us per 64 K bursts)
  if (!RAW && NSMAX == 148 && wide && (N & 1) == 0) {
    fused_demod<SPS, 64>(T, ph[wave], v, N, amp, toa, lane, sb, hb, nsoft, [] {}, nullptr, nullptr);
    return;
  }
  demod_core<SPS, RAW, NSMAX>(T, ph[wave], xb, N, wide, v, amp, toa, lane, sb, hb,
                              RAW ? reinterpret_cast<cx *>(soft) + (size_t)b * stride : nullptr, nsoft);
}

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
//   k_demod<1,RAW> : delayVector(burst/amp, -(TOA - chanOffset)), one wave per burst.
//   k_eq_dfe    : one lane per burst.  7-tap feed-forward FIR + the 156-step decision-feedback
//                 recursion of equalizeBurst (:1352-1384) and the slicer.
// ---------------------------------------------------------------------------------------------
#define EQ_NC 36            /* max correlation lags kept per burst */

// interpolatePoint on a per-lane LDS column of n entries (sigProcLib.cpp:639-659), ix on the 1/512 grid
__device__ __forceinline__ cx eq_interp(const TrxTables *__restrict__ T, const cx (*col)[64], int lane, int n, float ix) {
  const float fl = floorf(ix);
  const int I = (int)fl;
  const int f = (int)((ix - fl) * 512.0f) & 511;
  int start = I - 10;
  if (start < 0) start = 0;
  int end = I + 11;
  if ((unsigned)end > (unsigned)(n - 1)) end = n - 1;      // :646 (unsigned compare: negative end -> n-1)
  const float *row = T->sinc_grid[f];
  cx p = mk(0, 0);
  for (int i = start; i < end; i++) p = cadd(p, cmulr(col[i][lane], row[i - I + 10]));
  return p;
}

__global__ __launch_bounds__(64) void k_eq_detect(const TrxTables *__restrict__ T, const cx *__restrict__ samples,
                                                  const int32_t *__restrict__ offset,
                                                  const int32_t *__restrict__ length, int B, int tsc,
                                                  float detect_thresh, float energy_thresh, int variant52m,
                                                  int max_toa, uint8_t *__restrict__ flags,
                                                  cx *__restrict__ amp_out, float *__restrict__ toa_out,
                                                  float *__restrict__ toa_eq, cx *__restrict__ w_out,
                                                  cx *__restrict__ b_out, float snr_thresh, float snr_value,
                                                  float *__restrict__ chan_off_out) {
  // snr_value > 0: the SNR estimate itself (the Transceiver facade forms it on the host in the reference's
  // double arithmetic, Transceiver.cpp:340); else snr_thresh >= 0: the threshold that enters
  // SNR = |amp|^2/(thr^2+1); else energy_thresh.  chan_off_out (optional): chanRespOffset (:343).
  __shared__ cx corr[EQ_NC][64];
  __shared__ cx shf[EQ_NC][64];
  const int lane = threadIdx.x;
  const int b = blockIdx.x * 64 + lane;
  if (b >= B) return;                                      // no barriers below: each lane owns its columns
  const int off = offset[b], N = length[b];
  uint8_t fl = 0;
  cx amp = mk(0, 0);
  float toa = 0.0f;
  const bool good = (off >= 0) && (N >= 92) && (N <= 157);
  if (!good) {
    flags[b] = TRXSIG_F_BADLEN; amp_out[b] = amp; toa_out[b] = 0.0f; toa_eq[b] = 0.0f;
    return;
  }
  const cx *x = samples + off;

  // ---- energyDetect (:916-932; the 52M variant strides by 4, ref52:946-963) ----
  {
    float energy = 0.0f;
    const int step = variant52m ? 4 : 1;
    for (int i = 0; i < 20; i++) energy += norm2(x[i * step]);
    const bool ok = energy_thresh < 0.0f || energy / (float)20u > energy_thresh * energy_thresh;
    if (!ok) { flags[b] = 0; amp_out[b] = amp; toa_out[b] = 0.0f; toa_eq[b] = 0.0f; return; }
    fl = TRXSIG_F_ENERGY;
  }

  // ---- correlation ----
  int ncorr, winStart, La, startIndex;
  unsigned maxTOA = (unsigned)max_toa;
  if (!variant52m) {
    ncorr = 36; winStart = 56; La = 36; startIndex = 7;    // NO_DELAY, Lb = 16 (:951-955, 295-300)
  } else {                                                 // ref52:983-1000
    if (maxTOA < 3) maxTOA = 3;
    unsigned spanTOA = maxTOA;
    if (spanTOA < 5) spanTOA = 5;
    winStart = (int)(66 - spanTOA);
    La = (int)(16 + 2 * spanTOA);
    ncorr = (int)(2 * maxTOA + 1);
    const unsigned expectedTOAPeak = (unsigned)round((double)((T->mid_toa[tsc] + 5.0f) + (float)(size_t)((16 - 1) / 2)));
    startIndex = (int)(expectedTOAPeak - maxTOA);
    if (ncorr > EQ_NC || winStart < 0 || winStart + La > N) {
      flags[b] = TRXSIG_F_BADLEN; amp_out[b] = amp; toa_out[b] = 0.0f; toa_eq[b] = 0.0f;
      return;
    }
  }
  for (int i = 0; i < ncorr; i++) {
    const int t = startIndex + i;
    cx sum = mk(0, 0);
#pragma unroll
    for (int j = 0; j < 16; j++) {                         // tmp[j] = conj(mid[15-j]) (:480-498), j ascending
      const int ai = t - j;
      if (ai >= 0 && ai < La) sum = cadd(sum, cmul(x[winStart + ai], T->mid_ctap[tsc][15 - j]));
    }
    corr[i][lane] = sum;
  }

  // ---- peakDetect (:663-711) ----
  float maxP = 0.0f, maxIndex = -1.0f;
  for (int i = 0; i < ncorr; i++) {
    const float p = norm2(corr[i][lane]);
    if (p > maxP) { maxP = p; maxIndex = (float)i; }
  }
  float early = maxIndex - 1.0f, late = maxIndex + 1.0f, incr = 0.5f;
  for (int step = 0; step < 9; step++) {
    const cx e = eq_interp(T, corr, lane, ncorr, early), l = eq_interp(T, corr, lane, ncorr, late);
    const float ne = norm2(e), nl = norm2(l);
    if (ne < nl) early += incr;
    else if (ne > nl) early -= incr;
    else break;
    incr = incr * 0.5f;
    late = early + 2.0f;
  }
  toa = early + 1.0f;
  amp = eq_interp(T, corr, lane, ncorr, toa);

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
      if (p - i >= 0) { valley += norm2(corr[p - i][lane]); numRms++; }
      if (p + i < ncorr) { valley += norm2(corr[p + i][lane]); numRms++; }
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
        const cx (*src)[64] = corr;
        if (fabs((double)frac) > 1e-2) {
          const float *row = T->sinc_grid[(int)(frac * 512.0f) & 511];
          for (int t = 0; t < ncorr; t++) {
            cx sum = mk(0, 0);
            for (int j = 0; j < 21; j++) {
              const int ai = t + 10 - j;
              if (ai >= 0 && ai < ncorr) sum = cadd(sum, cmulr(corr[ai][lane], row[j]));
            }
            shf[t][lane] = sum;
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
  if (!detected) return;

  // ---- Transceiver.cpp:341-347: SNR, scaleVector(chan, 1/amp), designDFE(chan, SNR, 7) (:1246-1340) ----
  const float thr = snr_thresh >= 0.0f ? snr_thresh : (energy_thresh < 0.0f ? 0.0f : energy_thresh);
  const float snr = snr_value > 0.0f ? snr_value : (float)((double)norm2(amp) / ((double)(thr * thr) + 1.0));
  const cx ainv = cdiv(mk(1.0f, 0.0f), amp);
#pragma unroll
  for (int k = 0; k < 6; k++) chan[k] = cmul(chan[k], ainv);

  constexpr int Nf = 7, nu = 5;
  cx G0[Nf], G1[Nf];
#pragma unroll
  for (int k = 0; k < Nf; k++) { G0[k] = mk(0, 0); G1[k] = mk(0, 0); }
  G0[0] = mk((float)(1.0 / (double)sqrtf(snr)), 0.0f);     // :1261
#pragma unroll
  for (int j = 0; j <= nu; j++) G1[j] = mk(chan[j].r, -chan[j].i);
  cx Lu[Nf - 1][Nf - 1];                                   // L[i][j], i < j <= Nf-1, stored at [i][j-i-1]
  cx Lfb[nu];                                              // L[Nf-1][Nf .. Nf+nu-1]
  float d = 0.0f;
#pragma unroll
  for (int i = 0; i < Nf; i++) {
    d = norm2(G0[0]) + norm2(G1[0]);                       // :1272
    const cx g0c = mk(G0[0].r, -G0[0].i), g1c = mk(G1[0].r, -G1[0].i);
#pragma unroll
    for (int k = 1; k < Nf; k++) {                         // *Lptr = (G0[k]*conj(G0[0]) + G1[k]*conj(G1[0]))/d (:1277)
      const int col = i + k;
      const bool need = (i < Nf - 1) ? (col <= Nf - 1) : (col >= Nf && col < Nf + nu);
      if (need) {
        const cx tt = cadd(cmul(G0[k], g0c), cmul(G1[k], g1c));
        const cx v = mk(tt.r / d, tt.i / d);
        if (i < Nf - 1) Lu[i][k - 1] = v; else Lfb[col - Nf] = v;
      }
    }
    const cx kk = cdiv(G1[0], G0[0]);                      // :1282
    if (i != Nf - 1) {
      cx G0n[Nf], G1n[Nf];
      const cx kc = mk(kk.r, -kk.i), km = cmulr(kk, -1.0f);
#pragma unroll
      for (int q = 0; q < Nf; q++) G0n[q] = cadd(cmul(G1[q], kc), G0[q]);      // :1285-1287
#pragma unroll
      for (int q = 0; q < Nf; q++) G1n[q] = cadd(cmul(G0[q], km), G1[q]);      // :1289-1291
#pragma unroll
      for (int q = 0; q < Nf - 1; q++) G1n[q] = G1n[q + 1];                     // delayVector(G1new,-1) (:1292)
      G1n[Nf - 1] = mk(0, 0);
      const cx sc = mk((float)(1.0 / (double)sqrtf((float)(1.0 + (double)norm2(kk)))), 0.0f);   // :1294-1295
#pragma unroll
      for (int q = 0; q < Nf; q++) { G0[q] = cmul(G0n[q], sc); G1[q] = cmul(G1n[q], sc); }
    }
  }
  cx bq[nu];
#pragma unroll
  for (int j = 0; j < nu; j++) {                           // :1301-1304: * -1, conj
    const cx t1 = cmul(Lfb[j], mk(-1.0f, 0.0f));
    bq[j] = mk(t1.r, -t1.i);
  }
  cx v[Nf];
  v[Nf - 1] = mk(1.0f, 0.0f);
#pragma unroll
  for (int k = Nf - 2; k >= 0; k--) {                      // :1310-1319
    cx vk = mk(0, 0);
#pragma unroll
    for (int j = k + 1; j < Nf; j++) {
      const cx pr = cmul(v[j], Lu[k][j - k - 1]);
      vk.r -= pr.r; vk.i -= pr.i;
    }
    v[k] = vk;
  }
#pragma unroll
  for (int i = 0; i < Nf; i++) {                           // :1323-1335
    cx wi = mk(0, 0);
    const int endPt = (nu < (Nf - 1 - i)) ? nu : (Nf - 1 - i);
#pragma unroll
    for (int k = 0; k < Nf; k++)
      if (k < endPt + 1) wi = cadd(wi, cmul(v[i + k < Nf ? i + k : Nf - 1], mk(chan[k < 6 ? k : 5].r, -chan[k < 6 ? k : 5].i)));
    w_out[(size_t)b * Nf + i] = mk(wi.r / d, wi.i / d);
  }
#pragma unroll
  for (int j = 0; j < nu; j++) b_out[(size_t)b * nu + j] = bq[j];
}

// equalizeBurst after its delayVector: xd = delayed, scaled burst (B x xstride complex)
__global__ __launch_bounds__(64) void k_eq_dfe(const TrxTables *__restrict__ T, const cx *__restrict__ xd, int xstride,
                                               const int32_t *__restrict__ length, int B,
                                               const uint8_t *__restrict__ flags, const cx *__restrict__ w_in,
                                               const cx *__restrict__ b_in, float *__restrict__ soft,
                                               uint8_t *__restrict__ hard, int nsoft, int stride) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  float *sb = soft + (size_t)b * stride;
  uint8_t *hb = hard ? hard + (size_t)b * stride : nullptr;
  if (!(flags[b] & TRXSIG_F_DETECT)) {
    for (int m = 0; m < nsoft; m++) { sb[m] = 0.0f; if (hb) hb[m] = 0; }
    return;
  }
  const int N = length[b];
  const cx *x = xd + (size_t)b * xstride;
  cx w[7], bq[5], hist[5], win[7];
#pragma unroll
  for (int j = 0; j < 7; j++) w[j] = w_in[(size_t)b * 7 + j];
#pragma unroll
  for (int j = 0; j < 5; j++) { bq[j] = b_in[(size_t)b * 5 + j]; hist[j] = mk(0, 0); }
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
// k_modulate: modulateBurst (sigProcLib.cpp:521-565) + the scaleVector of Transceiver::addRadioVector
//   (Transceiver.cpp:108).  One workgroup per burst, one thread per output sample.
//
// out[t] = sum_{j=0..2*sps} a[t+sps-j]*p[j] (NO_DELAY, real pulse, j ascending) where a is the
// zero-stuffed, GMSK-rotated symbol train: a[n] = rot[n]*(2*bit-1) when n is a multiple of sps and
// n/sps < 148, else rot[n]*0 = +-0.  Only j = t mod sps (+sps, +2*sps) meet a non-zero a[n]; the
// other terms add +-0 and are skipped.
// ---------------------------------------------------------------------------------------------
template <int SPS>
__global__ __launch_bounds__(256) void k_modulate(const TrxTables *__restrict__ T,
                                                  const uint8_t *__restrict__ bits,
                                                  const int32_t *__restrict__ guard,
                                                  const float *__restrict__ gain, int B,
                                                  cx *__restrict__ out, const int32_t *__restrict__ out_off) {
  const int b = blockIdx.x;
  if (b >= B) return;
  __shared__ float sym[148];                               // 2*(bit&1)-1
  for (int i = threadIdx.x; i < 148; i += blockDim.x) sym[i] = (float)(2.0 * (bits[(size_t)b * 148 + i] & 0x01) - 1.0);
  __syncthreads();
  const int g = guard[b];
  if (g < 0 || g > 9) return;                              // 157*sps rotation entries (:215-216)
  const int N = SPS * (148 + g);
  cx *o = out + out_off[b];
  const bool scale = gain != nullptr;
  const float gv = scale ? gain[b] : 1.0f;
  for (int t = threadIdx.x; t < N; t += blockDim.x) {
    cx sum = mk(0, 0);
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const int j = (t % SPS) + q * SPS;                   // ascending j
      const int n = t + SPS - j;
      if (j <= 2 * SPS && n >= 0 && n < N && n / SPS < 148) {
        const cx a = cmulr(T->rot[n], sym[n / SPS]);       // GMSKRotate, realOnly (:235-239)
        sum = cadd(sum, cmulr(a, T->pulse[j]));            // convolve, b real (:345-353)
      }
    }
    if (scale) sum = cmul(sum, mk(gv, 0.0f));              // scaleVector(x, complex(g)) (:719-722)
    o[t] = sum;
  }
}

// ---------------------------------------------------------------------------------------------
// k_resample: polyphaseResampleVector (sigProcLib.cpp:1157-1210), S independent streams, one thread
//   per output sample; the reference's exact index walk and summation order (real LPF branch).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resample(const cx *__restrict__ in, int n, long long in_stride, int S,
                                                  int P, int Q, const float *__restrict__ lpf, int L,
                                                  cx *__restrict__ out, long long out_stride, int nout) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  const int s = blockIdx.y;
  if (o >= nout || s >= S) return;
  const cx *x = in + (size_t)s * in_stride;
  const int outputIx = o + (L - 1) / 2 / Q;                // :1177
  const int branch = (int)(((long long)outputIx * Q) % P);
  int inOff = (int)(((long long)outputIx * Q - branch) / P);
  int fi = branch;
  while (inOff >= n) { inOff--; fi += P; }                 // :1183-1186
  cx sum = mk(0, 0);
  while (inOff >= 0 && fi < L) {                           // :1196-1200
    sum = cadd(sum, cmulr(x[inOff], lpf[fi]));
    inOff--; fi += P;
  }
  out[(size_t)s * out_stride + o] = sum;
}

// RadioInterface::unUSRPifyVector / USRPifyVector (radioInterface.cpp:74-116)
__global__ __launch_bounds__(256) void k_unpack_i16(const short2 *__restrict__ iq, long long n, int swap,
                                                    cx *__restrict__ out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const short2 v = iq[i];
    out[i] = swap ? mk((float)v.y, (float)v.x) : mk((float)v.x, (float)v.y);
  }
}
// fp16 I/Q storage (BASELINE config 5): widening is exact, so every downstream result equals the
// float pipeline's on the same values
__global__ __launch_bounds__(256) void k_unpack_f16(const __half2 *__restrict__ iq, long long n, cx *__restrict__ out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float2 v = __half22float2(iq[i]);
    out[i] = mk(v.x, v.y);
  }
}
// gain != 1: scaleVector(x, gain) first (RadioInterface::pushBuffer, radioInterface.cpp:149: 13500.0).  The
// reference multiplies by the complex (gain, 0): x.r*gain - x.i*0 and x.r*0 + x.i*gain, which for finite
// samples equal x.r*gain and x.i*gain up to the sign of a zero, and the sign is lost in the cast.
__global__ __launch_bounds__(256) void k_pack_i16(const cx *__restrict__ in, long long n, float gain, short2 *__restrict__ iq) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    cx v = in[i];
    if (gain != 1.0f) v = mk(v.r * gain, v.i * gain);
    short2 o;
    o.x = (short)(int)v.r;                                 // (short)itr->real(): truncation toward zero
    o.y = (short)(int)v.i;
    iq[i] = o;
  }
}


// ---------------------------------------------------------------------------------------------
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

// demodulateBurst (k_demod's arithmetic) for one detected burst whose samples sit in registers: pair
// q = hl + LPB*i of v[] holds samples 2q, 2q+1.  P: the burst's LDS staging area (DemodGeom<SPS,148>::U
// entries); whatever it held before is dead.  LPB lanes per burst; lane hl writes soft bits OPL*hl .. +OPL-1.
// staged(): called once the samples are in LDS and v[] is dead (k_normal_quad starts the next
// burst's loads there, into the same registers).
// tp_pre / rv_pre (optional): the 21 delay-filter taps for this TOA and the lane's OPL reverse-rotation
// values, when the caller has fetched them ahead of time.
template <int SPS, int LPB, typename HOOK>
__device__ __forceinline__ void fused_demod(const TrxTables *__restrict__ T, cx *P, const float4 (&v)[(157 * SPS / 2 + LPB - 1) / LPB],
                                            int N, cx amp, float toa, int hl, float *sb, uint8_t *hbp, int nsoft, HOOK staged,
                                            const float *tp_pre, const cx *rv_pre) {
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
    if ((float)f == f512) {                                // on the 1/512 grid (always, after peakDetect)
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
  if (lo >= 0 && (N & 1) == 0 && (2 * LPB) % SPS == 0) {
    // common case: nothing falls off the front, pairs are whole.  Pair q = hl + LPB*i sits at positions
    // u = 2q + lo, u + 1; successive i move both by 2*LPB positions = 2*LPB/SPS entries of the same phase.
    const int ua = 2 * hl + lo, ub = ua + 1;
    cx *pa = P + (ua % SPS) * D::QLEN + ua / SPS;
    cx *pb = P + (ub % SPS) * D::QLEN + ub / SPS;
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
      if (n0 < N && u0 >= 0 && u0 < D::U) P[(u0 % SPS) * D::QLEN + u0 / SPS] = cmul(mk(v[i].x, v[i].y), inv);
      if (n0 + 1 < N && u1 >= 0 && u1 < D::U) P[(u1 % SPS) * D::QLEN + u1 / SPS] = cmul(mk(v[i].z, v[i].w), inv);
    }
  }
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

template <int SPS, int LPB>
__global__ __launch_bounds__(64 * TRX_FUSED_WAVES) void k_normal_fused(
    const TrxTables *__restrict__ T, const cx *__restrict__ samples, const int32_t *__restrict__ offset,
    const int32_t *__restrict__ length, int B, TapArg taps, cx gain_inv, float mid_toa, float detect_thresh,
    float energy_thresh, uint8_t *__restrict__ flags, cx *__restrict__ amp_out, float *__restrict__ toa_out,
    float *__restrict__ avgpwr_out, float *__restrict__ soft, uint8_t *__restrict__ hard, int nsoft, int stride) {
  typedef FusedGeom<SPS, LPB> G;
  typedef typename G::D D;
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
template <int SPS, unsigned TAPCLS, bool DEMOD>
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

  const cx meta = rec[(size_t)G::NS * Bpad + bb];
  const int M = __float_as_int(meta.r);
  const float energy = meta.i;
  const bool good = M != -2;
#pragma unroll
  for (int s0 = 0; s0 < G::NS; s0 += 8) {
    const int sl = s0 + r;
    if (sl < G::NS) {
      const cx v = rec[(size_t)sl * Bpad + bb];
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

}  // namespace

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
int trx_rec_slots(int sps) {
  switch (sps) {
    case 1: return CorrGeom<1>::NS + 1;
    case 2: return CorrGeom<2>::NS + 1;
    case 4: return CorrGeom<4>::NS + 1;
  }
  return 0;
}

// class of every tap (see TapPattern): 1 = real part exactly +-1, 2 = imaginary part exactly +-1, 0 = neither
static unsigned tap_classes(const TrxTables *hT, int tsc) {
  unsigned m = 0;
  for (int k = 0; k < 16; k++) {
    const trx_c32 a = hT->mid_ctap[tsc][k];
    const unsigned c = (a.r == 1.0f || a.r == -1.0f) ? 1u : ((a.i == 1.0f || a.i == -1.0f) ? 2u : 0u);
    m |= c << (2 * k);
  }
  return m;
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
  if (!(variant & 2)) {
    k_tsc_peak<S><<<dim3((B + 63) / 64), dim3(64), 0, st>>>(dT, rec, Bpad, B, tsc, detect_thresh, energy_thresh,
                                                            flags, amp, toa, avgpwr);
  } else {
    // gain.inv() (Complex.h:154-160) in the reference's float arithmetic; this file is built with -ffp-contract=off
    const trx_c32 g = hT->mid_gain[tsc];
    const float n = g.i * g.i + g.r * g.r;
    trx_c32 ginv; ginv.r = g.r / n; ginv.i = -g.i / n;
    k_tsc_peak8<S><<<dim3((B + 31) / 32), dim3(256), 0, st>>>(dT, rec, Bpad, B, ginv, hT->mid_toa[tsc], detect_thresh,
                                                              energy_thresh, flags, amp, toa, avgpwr);
  }
  if (prof) prof->end(TRXSIG_K_TSC_PEAK, st);
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

template <int S, int LPB>
static void launch_normal_fused(hipStream_t st, const TrxTables *dT, const TrxTables *hT, const trx_c32 *samples,
                                const int32_t *off, const int32_t *len, int B, int tsc, float detect_thresh,
                                float energy_thresh, uint8_t *flags, trx_c32 *amp, float *toa, float *avgpwr,
                                float *soft, uint8_t *hard, int nsoft, int stride, int generic_taps) {
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
    if (nsoft > 0) {
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
                                   int generic_taps, TrxProfiler *prof) {
  if (B <= 0) return hipSuccess;
  if (nsoft > 148 || (lanes_per_burst != 64 && lanes_per_burst != 32 && lanes_per_burst != 16)) return hipErrorInvalidValue;
  if (prof) prof->begin(TRXSIG_K_NORMAL_FUSED, st);
#define TRX_FUSED_CASE(S)                                                                                              \
  case S:                                                                                                              \
    if (lanes_per_burst == 64)                                                                                         \
      launch_normal_fused<S, 64>(st, dT, hT, samples, off, len, B, tsc, detect_thresh, energy_thresh, flags, amp, toa, \
                                 avgpwr, soft, hard, nsoft, stride, generic_taps);                                                   \
    else if (lanes_per_burst == 16)                                                                                    \
      launch_normal_fused<S, 16>(st, dT, hT, samples, off, len, B, tsc, detect_thresh, energy_thresh, flags, amp, toa, \
                                 avgpwr, soft, hard, nsoft, stride, generic_taps);                                                   \
    else                                                                                                               \
      launch_normal_fused<S, 32>(st, dT, hT, samples, off, len, B, tsc, detect_thresh, energy_thresh, flags, amp, toa, \
                                 avgpwr, soft, hard, nsoft, stride, generic_taps);                                                   \
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

int trx_rach_rec_floats(int sps) {            // floats per burst in the rach record (complex slots + valley)
  switch (sps) {
    case 1: return 2 * RachGeom<1>::CSLOTS + RachGeom<1>::NVAL;
    case 2: return 2 * RachGeom<2>::CSLOTS + RachGeom<2>::NVAL;
    case 4: return 2 * RachGeom<4>::CSLOTS + RachGeom<4>::NVAL;
  }
  return 0;
}

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

hipError_t trx_launch_rach_fast(hipStream_t st, int sps, const TrxTables *dT, const trx_c32 *samples,
                                const int32_t *off, const int32_t *len, int B, float detect_thresh,
                                float energy_thresh, uint8_t *flags, trx_c32 *amp, float *toa, float *avgpwr,
                                TrxProfiler *prof) {
  if (B <= 0) return hipSuccess;
  const dim3 grid(B), block(64);
  if (prof) prof->begin(TRXSIG_K_RACH_CORR, st);
  switch (sps) {
    case 1: k_rach_fast<1><<<grid, block, 0, st>>>(dT, samples, off, len, B, detect_thresh, energy_thresh, flags, amp, toa, avgpwr); break;
    case 2: k_rach_fast<2><<<grid, block, 0, st>>>(dT, samples, off, len, B, detect_thresh, energy_thresh, flags, amp, toa, avgpwr); break;
    case 4: k_rach_fast<4><<<grid, block, 0, st>>>(dT, samples, off, len, B, detect_thresh, energy_thresh, flags, amp, toa, avgpwr); break;
    default: return hipErrorInvalidValue;
  }
  if (prof) prof->end(TRXSIG_K_RACH_CORR, st);
  return hipGetLastError();
}

hipError_t trx_launch_rach_detect(hipStream_t st, int sps, const TrxTables *dT, const trx_c32 *samples,
                                  const int32_t *off, const int32_t *len, int B, float detect_thresh,
                                  float energy_thresh, float *ws, int Bpad, uint8_t *flags, trx_c32 *amp,
                                  float *toa, float *avgpwr, TrxProfiler *prof) {
  if (B <= 0) return hipSuccess;
  switch (sps) {
    case 1: launch_rach_detect<1>(st, dT, samples, off, len, B, detect_thresh, energy_thresh, ws, Bpad, flags, amp, toa, avgpwr, prof); break;
    case 2: launch_rach_detect<2>(st, dT, samples, off, len, B, detect_thresh, energy_thresh, ws, Bpad, flags, amp, toa, avgpwr, prof); break;
    case 4: launch_rach_detect<4>(st, dT, samples, off, len, B, detect_thresh, energy_thresh, ws, Bpad, flags, amp, toa, avgpwr, prof); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t trx_launch_demod(hipStream_t st, int sps, const TrxTables *dT, const trx_c32 *samples,
                            const int32_t *off, const int32_t *len, int B, const trx_c32 *amp,
                            const float *toa, const uint8_t *flags, int need_mask, float *soft,
                            uint8_t *hard, int nsoft, int stride, TrxProfiler *prof) {
  if (B <= 0) return hipSuccess;
  const dim3 grid((B + TRX_DEMOD_WAVES - 1) / TRX_DEMOD_WAVES), block(64 * TRX_DEMOD_WAVES);
  if (prof) prof->begin(TRXSIG_K_DEMOD, st);
#define TRX_DEMOD_CASE(S)                                                                                          \
  case S:                                                                                                          \
    if (nsoft <= 148)                                                                                              \
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

hipError_t trx_launch_modulate(hipStream_t st, int sps, const TrxTables *dT, const uint8_t *bits, const int32_t *guard,
                               const float *gain, int B, trx_c32 *out, const int32_t *out_off, TrxProfiler *prof) {
  if (B <= 0) return hipSuccess;
  if (prof) prof->begin(TRXSIG_K_MODULATE, st);
  switch (sps) {
    case 1: k_modulate<1><<<dim3(B), dim3(256), 0, st>>>(dT, bits, guard, gain, B, out, out_off); break;
    case 2: k_modulate<2><<<dim3(B), dim3(256), 0, st>>>(dT, bits, guard, gain, B, out, out_off); break;
    case 4: k_modulate<4><<<dim3(B), dim3(256), 0, st>>>(dT, bits, guard, gain, B, out, out_off); break;
    default: return hipErrorInvalidValue;
  }
  if (prof) prof->end(TRXSIG_K_MODULATE, st);
  return hipGetLastError();
}

hipError_t trx_launch_resample(hipStream_t st, const trx_c32 *in, int n, long long in_stride, int S, int P, int Q,
                               const float *lpf, int L, trx_c32 *out, long long out_stride, int nout,
                               TrxProfiler *prof) {
  if (S <= 0 || nout <= 0) return hipSuccess;
  if (prof) prof->begin(TRXSIG_K_RESAMPLE, st);
  k_resample<<<dim3((nout + 255) / 256, S), dim3(256), 0, st>>>(in, n, in_stride, S, P, Q, lpf, L, out, out_stride, nout);
  if (prof) prof->end(TRXSIG_K_RESAMPLE, st);
  return hipGetLastError();
}

hipError_t trx_launch_convert(hipStream_t st, int pack, const void *in, long long n, int swap, void *out,
                              TrxProfiler *prof, float gain) {
  if (n <= 0) return hipSuccess;
  long long blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (prof) prof->begin(TRXSIG_K_CONVERT, st);
  if (pack == 2) k_unpack_f16<<<dim3((unsigned)blocks), dim3(256), 0, st>>>((const __half2 *)in, n, (trx_c32 *)out);
  else if (pack) k_pack_i16<<<dim3((unsigned)blocks), dim3(256), 0, st>>>((const trx_c32 *)in, n, gain, (short2 *)out);
  else k_unpack_i16<<<dim3((unsigned)blocks), dim3(256), 0, st>>>((const short2 *)in, n, swap, (trx_c32 *)out);
  if (prof) prof->end(TRXSIG_K_CONVERT, st);
  return hipGetLastError();
}

hipError_t trx_launch_equalize(hipStream_t st, const TrxTables *dT, const trx_c32 *samples, const int32_t *off,
                               const int32_t *len, int B, int tsc, float detect_thresh, float energy_thresh,
                               int variant52m, int max_toa, uint8_t *flags, trx_c32 *amp, float *toa, float *toa_eq,
                               trx_c32 *w, trx_c32 *bq, trx_c32 *xd, int xstride, float *soft, uint8_t *hard,
                               int nsoft, int stride, TrxProfiler *prof) {
  if (B <= 0) return hipSuccess;
  if (prof) prof->begin(TRXSIG_K_EQUALIZE, st);
  k_eq_detect<<<dim3((B + 63) / 64), dim3(64), 0, st>>>(dT, samples, off, len, B, tsc, detect_thresh, energy_thresh,
                                                        variant52m, max_toa, flags, amp, toa, toa_eq, w, bq, -1.0f, 0.0f, nullptr);
  k_demod<1, true, 157><<<dim3((B + TRX_DEMOD_WAVES - 1) / TRX_DEMOD_WAVES), dim3(64 * TRX_DEMOD_WAVES), 0, st>>>(dT, samples, off, len, B, amp, toa_eq, flags,
                                                           TRXSIG_F_DETECT, (float *)xd, nullptr, 0, xstride);
  k_eq_dfe<<<dim3((B + 63) / 64), dim3(64), 0, st>>>(dT, xd, xstride, len, B, flags, w, bq, soft, hard, nsoft, stride);
  if (prof) prof->end(TRXSIG_K_EQUALIZE, st);
  return hipGetLastError();
}


// the two halves of trx_launch_equalize on their own (the Transceiver facade caches DFE taps per timeslot):
// channel estimate + designDFE only (energy gate off, explicit SNR threshold) ...
hipError_t trx_launch_estimate_dfe(hipStream_t st, const TrxTables *dT, const trx_c32 *samples, const int32_t *off,
                                   const int32_t *len, int B, int tsc, float detect_thresh, float snr_thresh,
                                   float snr_value, int variant52m, int max_toa, uint8_t *flags, trx_c32 *amp, float *toa,
                                   float *toa_eq, float *chan_off, trx_c32 *w, trx_c32 *bq, TrxProfiler *prof) {
  if (B <= 0) return hipSuccess;
  if (prof) prof->begin(TRXSIG_K_EQUALIZE, st);
  k_eq_detect<<<dim3((B + 63) / 64), dim3(64), 0, st>>>(dT, samples, off, len, B, tsc, detect_thresh, -1.0f, variant52m,
                                                        max_toa, flags, amp, toa, toa_eq, w, bq, snr_thresh, snr_value, chan_off);
  if (prof) prof->end(TRXSIG_K_EQUALIZE, st);
  return hipGetLastError();
}
// ... and scaleVector(burst, 1/amp) + equalizeBurst(burst, toa_eq, w, b) with caller-supplied taps (7 + 5 per burst)
hipError_t trx_launch_equalize_taps(hipStream_t st, const TrxTables *dT, const trx_c32 *samples, const int32_t *off,
                                    const int32_t *len, int B, const trx_c32 *amp, const float *toa_eq,
                                    const uint8_t *flags, const trx_c32 *w, const trx_c32 *bq, trx_c32 *xd, int xstride,
                                    float *soft, uint8_t *hard, int nsoft, int stride, TrxProfiler *prof) {
  if (B <= 0) return hipSuccess;
  if (prof) prof->begin(TRXSIG_K_EQUALIZE, st);
  k_demod<1, true, 157><<<dim3((B + TRX_DEMOD_WAVES - 1) / TRX_DEMOD_WAVES), dim3(64 * TRX_DEMOD_WAVES), 0, st>>>(
      dT, samples, off, len, B, amp, toa_eq, flags, TRXSIG_F_DETECT, (float *)xd, nullptr, 0, xstride);
  k_eq_dfe<<<dim3((B + 63) / 64), dim3(64), 0, st>>>(dT, xd, xstride, len, B, flags, w, bq, soft, hard, nsoft, stride);
  if (prof) prof->end(TRXSIG_K_EQUALIZE, st);
  return hipGetLastError();
}
