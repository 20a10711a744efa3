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

constexpr int kTile = 256;                                   // outputs per workgroup (one per thread, every carrier)
constexpr int kTapPitch = 33;                                // 32 taps per branch + 1 (odd pitch)

// grid (tiles, windows, wideband streams).  tw: [C][16] = conj(m_c[j]) = exp(-j theta_c j).
// LDS: X[xcap] the tile's span of the window as complex float (zeros outside the window), TP[(P/g) x 33] the taps by branch.
template <int C>
__global__ __launch_bounds__(256) void k_channelise16(TrxResampleArgs a, const float2 *__restrict__ tw, int xcap, int kt) {
  extern __shared__ __attribute__((aligned(16))) char ch_lds[];
  cx *X = reinterpret_cast<cx *>(ch_lds);
  float *TP = reinterpret_cast<float *>(ch_lds + sizeof(cx) * (size_t)xcap);
  const int w = blockIdx.y, s = blockIdx.z;
  const int D = (a.L - 1) / 2 / a.Q;                        // sigProcLib.cpp:1177
  const int g = a.tap_g, nbr = a.P / g;
  for (int e = threadIdx.x; e < nbr * 32; e += 256) {
    const int bi = e >> 5, k = e & 31, fi = bi * g + a.P * k;
    TP[bi * kTapPitch + k] = (k < kt && fi < a.L) ? a.lpf[fi] : 0.0f;   // (the reference's walk ends at the filter's end: zero taps)
  }
  const int o0 = a.o_skip + blockIdx.x * kTile;
  const long long oq_first = (long long)(o0 + D) * a.Q;
  const int lo = (int)(oq_first / a.P) - 31;                // first window sample any tap of this tile can meet
  {
    const short2 *raw = reinterpret_cast<const short2 *>(a.in) + (size_t)s * a.in_stride;
    const short2 *hist = a.hist + (size_t)s * a.hist_len;
    const int base = w * a.win_step - a.hist_len;           // raw index of the window's sample 0
    for (int i = threadIdx.x; i < xcap; i += 256) {
      const int idx = lo + i;
      cx v = mk(0, 0);
      if (idx >= 0 && idx < a.n) {                           // outside: "skip the tap" (:1183-1186, :1196) = a zero sample
        const int r = base + idx;
        const short2 q = r < 0 ? hist[a.hist_len + r] : raw[r];
        v = a.swap ? mk((float)q.y, (float)q.x) : mk((float)q.x, (float)q.y);   // unUSRPifyVector (radioInterface.cpp:108-109)
      }
      X[i] = v;
    }
  }
  __syncthreads();
  const int o = o0 + (int)threadIdx.x;
  if (o >= a.n_out) return;
  const long long oq = (long long)(o + D) * a.Q;
  const int branch = (int)(oq % a.P), inOff = (int)(oq / a.P);
  const float *tp = TP + (branch / g) * kTapPitch;
  const cx *xs = X + (inOff - lo);                          // xs[-k] = window sample inOff - k
  cx T[16];
#pragma unroll
  for (int j = 0; j < 16; j++) T[j] = mk(0, 0);
#pragma unroll
  for (int k = 0; k < 32; k++) {                            // taps past the filter's end are zeros
    const float h = tp[k];
    const cx x = xs[-k];
    T[k & 15].r = fma_af(h, x.r, T[k & 15].r);
    T[k & 15].i = fma_af(h, x.i, T[k & 15].i);
  }
  const int rho = inOff & 15;                               // the window starts on a multiple of 16 raw samples (the host checks)
  cx *out = reinterpret_cast<cx *>(a.out) + (size_t)w * a.out_win_step + (o - a.o_skip);
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
    const float2 m = tw[c * 16 + rho];                      // conj(m_c[rho]); y = m_c[rho] * acc
    cx y;
    y.r = fma_af(acc.r, m.x, acc.i * m.y);
    y.i = fma_af(acc.i, m.x, -(acc.r * m.y));
    out[(size_t)(s * C + c) * a.out_stride] = y;
  }
}

}  // namespace

// tw: device table [C][16] of exp(-j theta_c j).  Requirements (checked by the caller, trxsig_frontend.cpp): int16 input, window
// start and length multiples of 16 raw samples, at most 32 taps per output, C in {1, 2, 4, 8, 16}.
hipError_t trx_launch_channelise16(hipStream_t st, TrxResampleArgs a, int S_wide, int C, int n_windows, const float2 *tw, TrxProfiler *prof) {
  if (S_wide <= 0 || n_windows <= 0 || a.n_out <= a.o_skip) return hipSuccess;
  const int kt = (a.L + a.P - 1) / a.P;
  if (kt > 32 || S_wide > 65535 || n_windows > 65535) return hipErrorInvalidValue;
  int g = a.P, r = a.Q % a.P;
  while (r) { const int t = g % r; g = r; r = t; }          // gcd(P, Q): only branches that are multiples of it occur
  a.tap_g = g;
  const int xcap = (int)(((long long)(kTile - 1) * a.Q) / a.P + 32 + 4);
  const size_t lds = sizeof(trx_c32) * (size_t)xcap + sizeof(float) * (size_t)(a.P / g) * kTapPitch;
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  const dim3 grid((a.n_out - a.o_skip + kTile - 1) / kTile, n_windows, S_wide), block(256);
  if (prof) prof->begin(TRXSIG_K_RESAMPLE, st);
  switch (C) {
    case 1: k_channelise16<1><<<grid, block, lds, st>>>(a, tw, xcap, kt); break;
    case 2: k_channelise16<2><<<grid, block, lds, st>>>(a, tw, xcap, kt); break;
    case 4: k_channelise16<4><<<grid, block, lds, st>>>(a, tw, xcap, kt); break;
    case 8: k_channelise16<8><<<grid, block, lds, st>>>(a, tw, xcap, kt); break;
    case 16: k_channelise16<16><<<grid, block, lds, st>>>(a, tw, xcap, kt); break;
    default: if (prof) prof->end(TRXSIG_K_RESAMPLE, st); return hipErrorInvalidValue;
  }
  if (prof) prof->end(TRXSIG_K_RESAMPLE, st);
  return hipGetLastError();
}
