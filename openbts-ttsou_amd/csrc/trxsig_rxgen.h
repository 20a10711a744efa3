// trxsig_rxgen.h -- RadioInterface::pullBuffer's resampler (radioInterface.cpp:244-252: polyphaseResampleVector(65*sps : 96)
// of [192-sample history | 864-sample chunk], first 130*sps outputs dropped) evaluated ON DEMAND inside the burst kernels:
// a resampled sample is four multiply-adds on raw int16 samples, so the complex float32 stream (300 MB per 60 K bursts,
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
// burst s*nb + j of this call (launch index b itself, or a.sel[b] when a selection is given): 156 symbols, 157 when
// TN % 4 == 0 (radioInterface.cpp:370-378)
__device__ __forceinline__ RxBurst rx_burst(const TrxRxGen &a, int b) {
  RxBurst u;
  const int cell = a.sel ? a.sel[b] : b;
  u.s = cell / a.nb;
  const int j = cell - u.s * a.nb, tn = a.tn0 + j;
  const int long_before = (tn + 3) / 4 - (a.tn0 + 3) / 4;
  u.g0 = (156 * j + long_before) * 4 - a.tail;
  u.N = (156 + ((tn & 3) == 0)) * 4;
  return u;
}

// Where resampled sample g comes from: j0 = raw index of tap 0's sample, br = branch, n = the branch's slot in the tap table.
// 2340 outputs advance the walk by exactly 864 inputs (2340*96 = 864*260), so j0 and br run on across chunk boundaries and
// a fixed step of d samples is a fixed (d*96 % 260, d*96 / 260) step with a carry.
// Tap table: 96 = 4*24 and 260 = 4*65, so only the 65 branches 4 m ever occur and consecutive outputs visit m, m + 24,
// m + 48, ... (mod 65).  The table is stored in THAT order -- slot n holds branch 4*(24 n mod 65), i.e. n = 19 m mod 65 --
// so that consecutive outputs (consecutive lanes) read consecutive 16-byte slots: conflict-free, where indexing by branch
// put every lane of a wave on the same four LDS banks (96*4 dwords = 0 mod 64).
constexpr int RXG_NT = 65;
struct RxIdx { int j0, br, n; };
__device__ __forceinline__ RxIdx rx_index(int g, int skipD) {
  const unsigned u = (unsigned)(g + RXG_PC);               // g >= -2340
  const unsigned cq = u / RXG_PC, r = u - cq * RXG_PC;
  const unsigned oq = ((unsigned)skipD + r) * RXG_Q;
  const unsigned io = oq / RXG_P;
  RxIdx x;
  x.br = (int)(oq - io * RXG_P);
  x.j0 = RXG_CH * ((int)cq - 1) - RXG_HIST + (int)io;
  x.n = (int)((19u * ((unsigned)x.br >> 2)) % RXG_NT);
  return x;
}
template <int D>                                           // D samples further on, 0 < D < 2340
__device__ __forceinline__ RxIdx rx_step(RxIdx x) {
  constexpr int DB = (D * RXG_Q) % RXG_P, DI = (D * RXG_Q) / RXG_P, DN = D % RXG_NT;
  x.br += DB;
  const bool carry = x.br >= RXG_P;
  x.br -= carry ? RXG_P : 0;
  x.j0 += DI + (carry ? 1 : 0);
  x.n += DN;
  x.n -= x.n >= RXG_NT ? RXG_NT : 0;
  return x;
}
// The end-of-window rule (:1183-1186: a tap whose sample lies at or beyond the window's end is skipped) only concerns the
// last few outputs of a chunk: tap 0 for the w0 = 2340 - rl0 outputs before a chunk boundary, tap 1 for the last w1.  A
// burst holds at most one boundary; rx_boundary = how many samples after g0 it comes (1..2340).
__device__ __forceinline__ int rx_boundary(int g0) {
  const unsigned u = (unsigned)(g0 + RXG_PC);
  return RXG_PC - (int)(u % RXG_PC);
}

// raw sample j of stream s (j >= -1056): one unconditional load -- the previous push's kept window for j < 0, this push
// otherwise; an index past the push's last sample is clamped (such a sample only ever meets a tap the end-of-window rule
// has zeroed, and it is a finite int16 value either way).  rx_widen: the float pair unUSRPifyVector makes of it
// (radioInterface.cpp:91-116), applied once all of a wave's loads are in flight.
__device__ __forceinline__ short2 rx_raw(const TrxRxGen &a, int s, int j) {
  const int top = a.K * RXG_CH - 1;
  const int jc = j > top ? top : j;
  const short2 *p = jc < 0 ? a.keep + ((size_t)s * RXG_NIN + (RXG_NIN + jc)) : a.raw + ((size_t)s * a.raw_stride + jc);
  return *p;
}
__device__ __forceinline__ cx rx_widen(const TrxRxGen &a, short2 v) {
  return a.swap ? mk((float)v.y, (float)v.x) : mk((float)v.x, (float)v.y);
}

// X[t] = raw sample jlo + t (LDS, staged by the caller), TPB = the tap table (LDS, slot order); d = samples from this one to
// the next chunk boundary.  The reference skips a tap whose sample lies at or beyond the window's end; here that tap is made
// zero instead, the product is +-0, and since the sum starts at +0 and +0 + (+-0) = +0 the running sum is the reference's
// bit for bit (the sample read in its place is a finite int16 value).  No branch, no conditional load.
__device__ __forceinline__ cx rx_sample(const cx *X, const float4 *TPB, int jlo, RxIdx ix, int d, int w0, int w1) {
  const cx *x = X + (ix.j0 - jlo);
  float4 tp = TPB[ix.n];
  tp.x = (unsigned)(d - 1) < (unsigned)w0 ? 0.0f : tp.x;
  tp.y = (unsigned)(d - 1) < (unsigned)w1 ? 0.0f : tp.y;
  cx sum = mk(0, 0);
  sum = cadd(sum, cmulr(x[0], tp.x));
  sum = cadd(sum, cmulr(x[-1], tp.y));
  sum = cadd(sum, cmulr(x[-2], tp.z));
  sum = cadd(sum, cmulr(x[-3], tp.w));
  return sum;
}

// Resampled sample n of a burst straight from global memory (no staged stretch): four 4-byte loads that neighbouring lanes
// share through the vector cache.  For kernels whose LDS is spoken for (the access-burst detector).
struct RxGlobalSrc {
  TrxRxGen a;
  RxBurst u;
  int nb;                                                  // samples from the burst's start to the next chunk boundary
  __device__ __forceinline__ RxGlobalSrc(const TrxRxGen &gen, int b) : a(gen), u(rx_burst(gen, b)), nb(rx_boundary(u.g0)) {}
  __device__ __forceinline__ cx at(RxIdx ix, int n) const {
    float4 tp = a.tpb[ix.n];
    const int d = nb - n;
    tp.x = (unsigned)(d - 1) < (unsigned)a.w0 ? 0.0f : tp.x;
    tp.y = (unsigned)(d - 1) < (unsigned)a.w1 ? 0.0f : tp.y;
    const short2 r0 = rx_raw(a, u.s, ix.j0), r1 = rx_raw(a, u.s, ix.j0 - 1), r2 = rx_raw(a, u.s, ix.j0 - 2), r3 = rx_raw(a, u.s, ix.j0 - 3);
    cx sum = mk(0, 0);
    sum = cadd(sum, cmulr(rx_widen(a, r0), tp.x));
    sum = cadd(sum, cmulr(rx_widen(a, r1), tp.y));
    sum = cadd(sum, cmulr(rx_widen(a, r2), tp.z));
    sum = cadd(sum, cmulr(rx_widen(a, r3), tp.w));
    return sum;
  }
  __device__ __forceinline__ RxIdx index(int n) const { return rx_index(u.g0 + n, a.skipD); }
};

}  // namespace
