// trxsig_rxgen.h -- RadioInterface::pullBuffer's resampler (radioInterface.cpp:244-252: polyphaseResampleVector(65*sps : 96)
// of [192-sample history | 864-sample chunk], first 130*sps outputs dropped) evaluated ON DEMAND inside the burst kernels:
// a resampled sample is four multiply-adds on raw int16 samples, so the complex float32 stream (300 MB per 61 K bursts,
// written once and read back 1.4 times by the unfused chain) never exists in HBM.  sps = 4 only (P = 260, L <= 4 P).
//
// Resampled sample g (counted from the first sample this push produces; g < 0 = the previous push's last chunk):
//   chunk c = floor(g / 2340), r = g - 2340 c, outputIx = 520 + r; the reference's walk (sigProcLib.cpp:1177-1200) starts at
//   outputIx*Q + (L-1)/2/Q*Q = branch + 260*inOff and takes taps branch + 260 k against window samples inOff - k, k = 0..3,
//   skipping samples at or beyond the window's end (1056) -- never before its start, since inOff >= 193 for a kept output.
//   Window sample i of chunk c is raw sample 864 c - 192 + i of the stream (raw index < 0: the kept window of the last push).
// Same terms in the same order as k_rx_resample (tests/test_gpu_config4.py compares the fused calls with the unfused chain).
#pragma once
#include "trxsig_dev.h"

namespace {

constexpr int RXG_PC = 2340, RXG_P = 260, RXG_Q = 96, RXG_NIN = 1056, RXG_CH = 864, RXG_HIST = 192;

struct RxBurst { int s, g0, N; };
// burst b = s*nb + j of this call: 156 symbols, 157 when TN % 4 == 0 (radioInterface.cpp:370-378)
__device__ __forceinline__ RxBurst rx_burst(const TrxRxGen &a, int b) {
  RxBurst u;
  u.s = b / a.nb;
  const int j = b - u.s * a.nb, tn = a.tn0 + j;
  const int long_before = (tn + 3) / 4 - (a.tn0 + 3) / 4;
  u.g0 = (156 * j + long_before) * 4 - a.tail;
  u.N = (156 + ((tn & 3) == 0)) * 4;
  return u;
}

struct RxIdx { int j0, br, io; };                          // raw index of tap 0, branch, window-relative index of tap 0
__device__ __forceinline__ RxIdx rx_index(int g, int skipD) {
  const unsigned u = (unsigned)(g + RXG_PC);               // g >= -2340
  const unsigned cq = u / RXG_PC, r = u - cq * RXG_PC;
  const unsigned oq = ((unsigned)skipD + r) * RXG_Q;
  RxIdx x;
  x.io = (int)(oq / RXG_P);
  x.br = (int)(oq - (unsigned)x.io * RXG_P);
  x.j0 = RXG_CH * ((int)cq - 1) - RXG_HIST + x.io;
  return x;
}

// raw sample j of stream s as the float pair unUSRPifyVector makes of it (radioInterface.cpp:91-116)
__device__ __forceinline__ cx rx_raw(const TrxRxGen &a, int s, int j) {
  short2 v = make_short2(0, 0);
  if (j < 0) v = a.keep[(size_t)s * RXG_NIN + (RXG_NIN + j)];
  else if (j < a.K * RXG_CH) v = a.raw[(size_t)s * a.raw_stride + j];
  return a.swap ? mk((float)v.y, (float)v.x) : mk((float)v.x, (float)v.y);
}

// X[t] = raw sample jlo + t (LDS, staged by the caller), TPB = the branch-major taps (LDS)
__device__ __forceinline__ cx rx_sample(const cx *X, const float4 *TPB, int jlo, RxIdx ix) {
  const int t = ix.j0 - jlo;
  const float4 tp = TPB[ix.br];
  const cx z = mk(0, 0);
  const cx x0 = ix.io < RXG_NIN ? X[t] : z, x1 = ix.io - 1 < RXG_NIN ? X[t - 1] : z, x2 = X[t - 2], x3 = X[t - 3];
  cx sum = mk(0, 0);
  sum = cadd(sum, cmulr(x0, tp.x));
  sum = cadd(sum, cmulr(x1, tp.y));
  sum = cadd(sum, cmulr(x2, tp.z));
  sum = cadd(sum, cmulr(x3, tp.w));
  return sum;
}

}  // namespace
