// trxsig_chan.hip -- the multi-ARFCN channeliser in its SHARED-FILTER form (round 4; SURVEY 8f rank 4, a component the
// reference does not have): C carriers on a grid of sixteenths of the wideband sample rate leave one wideband int16 stream
// through ONE pass over the samples.
//
// The per-carrier form (k_resample<int16, mix>, trxsig_tx.hip: bit-equal to the reference's frequencyShift +
// polyphaseResampleVector per carrier) computes, for carrier c and output o,
//     y_c[o] = sum_k h[branch + P k] * x[n - k] * m_c[n - k],   n = inOff(o),   m_c[i] = exp(j theta_c i)
// -- C times the 31-tap filter and C mixed copies of the window in LDS (0.73 ms per 12,000 bursts' worth: LDS-bound).  With
// theta_c a multiple of 2 pi / 16 the mixer has period 16, m_c[n - k] = m_c[n] * conj(m_c[k mod 16]), so
//     T_j   = sum_{k = j mod 16} h[branch + P k] * x[n - k]          (j = 0..15: real taps on raw samples, shared by every carrier)
//     y_c[o] = m_c[n mod 16] * sum_j conj(m_c[j]) * T_j             (sixteen complex multiply-adds per carrier, from registers)
// -- the window is staged once, unmixed; a sample and a tap are read once per output instead of C times; the mixer is two
// small constant tables.  Same sum up to the order of its terms: NOT bit-equal to the per-carrier form (which stays the pinned
// one and the default); checked against it at 1e-4 of the signal's scale with identical hard bits (tests/test_gpu_channeliser.py).
// Because the form is approximate by construction its multiply-adds may fuse (explicit v_fma_f32 with the marker comment
// "approx-form", counted apart by tools/asm_stats.py); the twiddles are exact cos / sin (double, rounded once).
// With four carriers or more the sixteen sums sum_j conj(m_c[j]) T_j are taken from ONE 16-point FFT of T (they are bins k_c of its
// DFT: theta_c = 2 pi k_c / 16) -- ~190 operations for all sixteen bins where the direct form spends 64 per carrier.
#include <cstdlib>
#include "trxsig_dev.h"

namespace {

__device__ __forceinline__ float fma_af(float a, float b, float c) {   // vector x vector
  float r;
  asm("v_fma_f32 %0, %1, %2, %3 ; approx-form" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float fma_as(float a, float b_sgpr, float c) {   // a uniform factor from an SGPR
  float r;
  asm("v_fma_f32 %0, %1, %2, %3 ; approx-form" : "=v"(r) : "v"(a), "s"(b_sgpr), "v"(c));
  return r;
}

// X[k] = sum_j a[j] exp(-2 pi i j k / 16), k = 0..15, in place; X[k] is left in a[bitrev4(k)] (decimation in frequency, radix 2)
__device__ __forceinline__ cx tw16(cx x, int m) {            // x * exp(-2 pi i m / 16), m a compile-time constant after unrolling
  constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, r = 0.70710678118654752f;
  switch (m & 15) {
    case 0: return x;
    case 1: return mk(x.r * c1 + x.i * s1, x.i * c1 - x.r * s1);
    case 2: return mk((x.r + x.i) * r, (x.i - x.r) * r);
    case 3: return mk(x.r * s1 + x.i * c1, x.i * s1 - x.r * c1);
    case 4: return mk(x.i, -x.r);
    case 5: return mk(x.i * c1 - x.r * s1, -(x.r * c1 + x.i * s1));
    case 6: return mk((x.i - x.r) * r, -((x.r + x.i) * r));
    case 7: return mk(x.i * s1 - x.r * c1, -(x.r * s1 + x.i * c1));
    default: return tw16(mk(-x.r, -x.i), m - 8);
  }
}
template <int SPAN>
__device__ __forceinline__ void fft16_stage(cx (&a)[16]) {    // (a template per stage: every index a compile-time constant, the array stays in registers)
#pragma unroll
  for (int o = 0; o < 16; o += 2 * SPAN) {
#pragma unroll
    for (int i = 0; i < SPAN; i++) {
      const cx u = a[o + i], v = a[o + i + SPAN];
      a[o + i] = mk(u.r + v.r, u.i + v.i);
      a[o + i + SPAN] = tw16(mk(u.r - v.r, u.i - v.i), i * (8 / SPAN));
    }
  }
}
__device__ __forceinline__ void fft16(cx (&a)[16]) {
  fft16_stage<8>(a); fft16_stage<4>(a); fft16_stage<2>(a); fft16_stage<1>(a);
}

constexpr int kTile = 256;                                   // outputs per workgroup (one per thread, every carrier)
constexpr int kTapPitch = 33;                                // 32 taps per branch + 1 (odd pitch)

// grid (tiles, windows, wideband streams).  tw: [C][16] = conj(m_c[j]) = exp(-j theta_c j).
// LDS: X[xcap] the tile's span of the window as complex float (zeros outside the window), TP[(P/g) x 33] the taps by branch.
template <int C>
__global__ __launch_bounds__(256) void k_channelise16(TrxResampleArgs a, const float2 *__restrict__ tw, int xcap, int kt, unsigned long long binmap, int tpw) {
  extern __shared__ __attribute__((aligned(16))) char ch_lds[];
  // the window as received (int16 pairs: a wave's 4-byte reads are one LDS pass, its 8-byte reads of converted samples were two to four;
  // the conversion is exact either way), the taps' rows in the order the outputs visit the branches (see k_rx_resample, trxsig_tx.hip)
  short2 *X = reinterpret_cast<short2 *>(ch_lds);
  float *TP = reinterpret_cast<float *>(ch_lds + ((sizeof(short2) * (size_t)xcap + 15) & ~(size_t)15));
  const int w = blockIdx.y, s = blockIdx.z;
  const int D = (a.L - 1) / 2 / a.Q;                        // sigProcLib.cpp:1177
  const int g = a.tap_g, nbr = a.P / g;
  // the rotation factors conj(m_c[rho]) too: rho differs from lane to lane, and a global load per carrier and output, each waited for
  // before its product, was where the kernel spent three quarters of its time (SQ_WAIT_ANY 74 %)
  float2 *TWS = reinterpret_cast<float2 *>(TP + (((size_t)nbr * kTapPitch + 1) & ~(size_t)1));
  for (int e = threadIdx.x; e < C * 16; e += 256) TWS[e] = tw[e];
  // (every load below is unconditional, its index clamped and the value selected afterwards: a load under a branch whose value is used
  //  inside the branch is waited for before the next one is issued -- four trips to L2 in a row per tile, eight for the taps)
  for (int e0 = 0; e0 < nbr * 32; e0 += 256 * 4) {
    float tv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int e = e0 + 256 * u + (int)threadIdx.x, bi = e >> 5, k = e & 31, fi = bi * g + a.P * k;
      tv[u] = a.lpf[fi < a.L ? fi : a.L - 1];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int e = e0 + 256 * u + (int)threadIdx.x, bi = e >> 5, k = e & 31, fi = bi * g + a.P * k;
      if (e < nbr * 32) {
        const int row = a.row_inv ? (int)(((unsigned)bi * (unsigned)a.row_inv) % (unsigned)nbr) : bi;
        TP[row * kTapPitch + k] = (k < kt && fi < a.L) ? tv[u] : 0.0f;   // (the reference's walk ends at the filter's end: zero taps)
      }
    }
  }
  // A workgroup serves `tpw` consecutive tiles with ONE staging of the taps, and the next tile's raw samples are loaded (into registers)
  // before the current tile is filtered: a one-tile workgroup lived for two trips to L2 and a few hundred instructions, and the kernel
  // was bound by how fast such workgroups can be turned over (167 us for 28 us' worth of instructions).
  constexpr int NQ = 4;                                     // NQ * 256 >= xcap (the launcher checks)
  const short2 *raw = reinterpret_cast<const short2 *>(a.in) + (size_t)s * a.in_stride;
  const short2 *hist = a.hist + (size_t)s * a.hist_len;
  const int base = w * a.win_step - a.hist_len;             // raw index of the window's sample 0
  const int n_tiles = (a.n_out - a.o_skip + kTile - 1) / kTile;
  const int t0 = blockIdx.x * tpw, t1 = t0 + tpw < n_tiles ? t0 + tpw : n_tiles;
  auto tile_lo = [&](int t) {                               // first window sample any tap of tile t can meet
    const long long oq_first = (long long)(a.o_skip + t * kTile + D) * a.Q;
    return (int)(oq_first / a.P) - 31;
  };
  short2 v[NQ];
  auto fetch = [&](int t) {
    const int lo = tile_lo(t);
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int i = (int)threadIdx.x + 256 * q, idx = lo + i;
      const bool in = i < xcap && idx >= 0 && idx < a.n;     // outside: "skip the tap" (:1183-1186, :1196) = a zero sample
      const int r = base + (in ? idx : 0);
      const short2 *p = r < 0 ? hist + (a.hist_len + r) : raw + r;
      const short2 qv = *p;                                 // (base + 0 is a valid sample of the stream or its history)
      v[q] = in ? qv : make_short2(0, 0);
    }
  };
  fetch(t0);
  for (int t = t0; t < t1; t++) {
  const int o0 = a.o_skip + t * kTile;
  const int lo = tile_lo(t);
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    const int i = (int)threadIdx.x + 256 * q;
    if (i < xcap) X[i] = a.swap ? make_short2(v[q].y, v[q].x) : v[q];   // unUSRPifyVector's order (radioInterface.cpp:108-109); its int16 -> float on the way out
  }
  lds_barrier();                                            // (X and, the first time, the taps: LDS only -- see lds_barrier)
  if (t + 1 < t1) fetch(t + 1);
  const int o = o0 + (int)threadIdx.x;
  if (o < a.n_out) {
  const long long oq = (long long)(o + D) * a.Q;
  const int branch = (int)(oq % a.P), inOff = (int)(oq / a.P);
  const int bi = branch / g;
  const float *tp = TP + (a.row_inv ? (int)(((unsigned)bi * (unsigned)a.row_inv) % (unsigned)nbr) : bi) * kTapPitch;
  const short2 *xs = X + (inOff - lo);                      // xs[-k] = window sample inOff - k
  cx T[16];
#pragma unroll
  for (int j = 0; j < 16; j++) T[j] = mk(0, 0);
#pragma unroll
  for (int k = 0; k < 32; k++) {                            // taps past the filter's end are zeros
    const float h = tp[k];
    const short2 xq = xs[-k];
    const cx x = mk((float)xq.x, (float)xq.y);
    T[k & 15].r = fma_af(h, x.r, T[k & 15].r);
    T[k & 15].i = fma_af(h, x.i, T[k & 15].i);
  }
  const int rho = inOff & 15;                               // the window starts on a multiple of 16 raw samples (the host checks)
  cx *out = reinterpret_cast<cx *>(a.out) + (size_t)w * a.out_win_step + (o - a.o_skip);
  if constexpr (C >= 4) {
    // binmap: four bits per carrier, k_c.  Bin by bin (a compile-time register), the carriers that sit on it (a uniform test)
    // binmap: four bits per carrier, k_c (no two carriers on one bin: the caller checks).  Bin by bin -- a compile-time register --: is there
    // a carrier on it (sixteen uniform tests), which one (its index only enters addresses)
    fft16(T);
    unsigned occupied = 0;
    unsigned long long who = 0;                              // four bits per bin: the carrier on it
#pragma unroll
    for (int c = 0; c < C; c++) {
      const unsigned kc = (unsigned)((binmap >> (4 * c)) & 15ull);
      occupied |= 1u << kc;
      who |= (unsigned long long)c << (4 * kc);
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {                          // T[i] = X[k], k = bitrev4(i)
      constexpr int dummy = 0; (void)dummy;
      const int k = ((i & 1) << 3) | ((i & 2) << 1) | ((i & 4) >> 1) | ((i & 8) >> 3);
      if ((occupied >> k) & 1u) {
        const int c = (int)((who >> (4 * k)) & 15ull);
        const cx acc = T[i];
        const float2 m = TWS[c * 16 + rho];                  // conj(m_c[rho]); y = m_c[rho] * acc
        cx y;
        y.r = fma_af(acc.r, m.x, acc.i * m.y);
        y.i = fma_af(acc.i, m.x, -(acc.r * m.y));
        out[(size_t)(s * C + c) * a.out_stride] = y;
      }
    }
  } else {
#pragma unroll
  for (int c = 0; c < C; c++) {
    cx acc = mk(0, 0);
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const float2 t = tw[c * 16 + j];                       // (uniform: scalar loads)
      acc.r = fma_as(T[j].r, t.x, acc.r);
      acc.r = fma_as(T[j].i, -t.y, acc.r);
      acc.i = fma_as(T[j].i, t.x, acc.i);
      acc.i = fma_as(T[j].r, t.y, acc.i);
    }
    const float2 m = TWS[c * 16 + rho];                      // conj(m_c[rho]); y = m_c[rho] * acc
    cx y;
    y.r = fma_af(acc.r, m.x, acc.i * m.y);
    y.i = fma_af(acc.i, m.x, -(acc.r * m.y));
    out[(size_t)(s * C + c) * a.out_stride] = y;
  }
  }
  }                                                         // (o < n_out)
  lds_barrier();                                            // every thread is done with X before the next tile lands (its stores may still be in flight)
  }                                                         // (tiles)
}

}  // namespace

// tw: device table [C][16] of exp(-j theta_c j).  Requirements (checked by the caller, trxsig_frontend.cpp): int16 input, window
// start and length multiples of 16 raw samples, at most 32 taps per output, C in {1, 2, 4, 8, 16}.
hipError_t trx_launch_channelise16(hipStream_t st, TrxResampleArgs a, int S_wide, int C, int n_windows, const float2 *tw, TrxProfiler *prof,
                                   unsigned long long binmap) {
  if (S_wide <= 0 || n_windows <= 0 || a.n_out <= a.o_skip) return hipSuccess;
  const int kt = (a.L + a.P - 1) / a.P;
  if (kt > 32 || S_wide > 65535 || n_windows > 65535) return hipErrorInvalidValue;
  int g = a.P, r = a.Q % a.P;
  while (r) { const int t = g % r; g = r; r = t; }          // gcd(P, Q): only branches that are multiples of it occur
  a.tap_g = g;
  const int xcap = (int)(((long long)(kTile - 1) * a.Q) / a.P + 32 + 4);
  const size_t lds = ((sizeof(short2) * (size_t)xcap + 15) & ~(size_t)15) + sizeof(float) * ((((size_t)(a.P / g) * kTapPitch) + 1) & ~(size_t)1) +
                     sizeof(float2) * 16 * (size_t)C;
  a.row_inv = 0;                                            // ((Q mod P) / g)^-1 mod P / g: the rows of the tap table in visiting order
  {
    const int nbr = a.P / g, st1 = (a.Q % a.P) / g;
    for (int v = 1; v < nbr; v++)
      if (((long long)v * st1) % nbr == 1) { a.row_inv = v; break; }
  }
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  if (xcap > 4 * 256) return hipErrorInvalidValue;
  const int n_tiles = (a.n_out - a.o_skip + kTile - 1) / kTile;
  int tpw = 1;                                              // tiles per workgroup: up to four (measured: 1: 174, 2: 130, 4: 117, 8: 145 us), as long as
  for (int cand = 4; cand > 1; cand >>= 1)                  // the machine keeps eight workgroups per CU
    if (cand <= n_tiles && (long long)S_wide * n_windows * ((n_tiles + cand - 1) / cand) >= 2048) { tpw = cand; break; }
  { const int v = trx_knob(TRX_KNOB_CHAN_TPW); if (v >= 1 && v <= 64) tpw = v; }   // (TRXSIG_TUNE_CHAN_TPW, A/B)
  const dim3 grid((n_tiles + tpw - 1) / tpw, n_windows, S_wide), block(256);
  if (prof) prof->begin(TRXSIG_K_RESAMPLE, st);
  switch (C) {
    case 1: k_channelise16<1><<<grid, block, lds, st>>>(a, tw, xcap, kt, binmap, tpw); break;
    case 2: k_channelise16<2><<<grid, block, lds, st>>>(a, tw, xcap, kt, binmap, tpw); break;
    case 4: k_channelise16<4><<<grid, block, lds, st>>>(a, tw, xcap, kt, binmap, tpw); break;
    case 8: k_channelise16<8><<<grid, block, lds, st>>>(a, tw, xcap, kt, binmap, tpw); break;
    case 16: k_channelise16<16><<<grid, block, lds, st>>>(a, tw, xcap, kt, binmap, tpw); break;
    default: if (prof) prof->end(TRXSIG_K_RESAMPLE, st); return hipErrorInvalidValue;
  }
  if (prof) prof->end(TRXSIG_K_RESAMPLE, st);
  return hipGetLastError();
}
