// trxsig_tx.hip -- transmit side and radio-format kernels: modulateBurst, polyphaseResampleVector,
// int16 / fp16 I/Q conversion.
// Numerical contract: see trxsig_dev.h / DESIGN.md (every float32 operation is the reference's, in the
// reference's order; built with -ffp-contract=off).
#include <cstdlib>

#include "trxsig_dev.h"

namespace {

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
  auto sample = [&](int t) {
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
    return sum;
  };
  if ((out_off[b] & 1) == 0) {                             // two samples per lane and one 16-byte store
    for (int u = threadIdx.x; 2 * u + 1 < N; u += blockDim.x) {
      const cx s0 = sample(2 * u), s1 = sample(2 * u + 1);
      *reinterpret_cast<float4 *>(o + 2 * u) = make_float4(s0.r, s0.i, s1.r, s1.i);
    }
    if ((N & 1) && threadIdx.x == 0) o[N - 1] = sample(N - 1);
  } else {
    for (int t = threadIdx.x; t < N; t += blockDim.x) o[t] = sample(t);
  }
}

// ---------------------------------------------------------------------------------------------
// k_resample: polyphaseResampleVector (sigProcLib.cpp:1157-1210; real LPF branch) for S independent streams, tiled:
//   a workgroup produces OB consecutive outputs of one input window.  The input span those outputs can touch and the
//   LPF taps are staged in LDS with coalesced loads (the span is re-read ~L/P times per sample and the taps are walked
//   at stride P -- both from LDS, not from HBM); every output is the reference's exact index walk and summation order.
//   IN_I16:  the input is the radio's int16 I/Q stream and unUSRPifyVector (radioInterface.cpp:91-116) happens on the way
//            into LDS; the stream is cut into windows of n samples that start win_step apart and reach hist_len samples
//            back (RadioInterface::pullBuffer's [history | chunk], :244-246), samples before the stream's start come
//            from `hist`; outputs [o_skip, n_out) of each window are kept (:249-252).
//   OUT_I16: scaleVector(gain) and USRPifyVector (:148-151, :74-89) on the way out (RadioInterface::pushBuffer).
// ---------------------------------------------------------------------------------------------
#define TRX_RES_XCAP 4096                                  // staged input samples per workgroup
#define TRX_RES_LCAP 4096                                  // staged taps

//   IN_BITS: (the transmit back end, fused) there is no input stream at all: the window is the concatenation of modulated
//            bursts, and a staged sample is computed from the burst's 148 bits -- modulateBurst's at most three non-zero
//            terms (k_modulate's arithmetic) and addRadioVector's gain -- so the complex float32 send buffer never exists.
//            a.tx: the bursts that overlap the window (start position, ring slot, guard, gain flag), a.in: the bit ring.
// Window samples [lo, hi] that bursts cover, computed from the bursts' bits: burst by burst, a thread per symbol period.
// Sample t = SPS u + r of a burst is modulateBurst's sum over j ascending of a[t + SPS - j] p[j] where only j = r, r + SPS
// (and r + 2 SPS for r = 0) meet a non-zero a[n] = rot[n] * (2 bit - 1), n = SPS (u + 1 - q) (k_modulate's arithmetic): the
// three symbols u + 1, u, u - 1 serve the period's SPS samples.
template <int SPS>
__device__ __forceinline__ void tx_stage_tile(const TrxTables *__restrict__ T, const uint8_t *__restrict__ ring, const float *__restrict__ gring,
                                              const int *tb_start, const int *tb_meta, int M, int lo, int hi, cx *X) {
  float pul[2 * SPS + 1];
#pragma unroll
  for (int j = 0; j < 2 * SPS + 1; j++) pul[j] = T->pulse[j];
  for (int m = 0; m < M; m++) {                            // (uniform: every thread walks the tile's bursts)
    const int start = tb_start[m];
    if (start > hi) break;
    const int meta = tb_meta[m], slot = meta & 0xffff, guard = (meta >> 16) & 0xf;
    const bool scale = (meta >> 20) & 1;
    const int nsym = 148 + guard;
    if (start + SPS * nsym <= lo) continue;
    const float gv = scale ? gring[slot] : 1.0f;
    const uint8_t *bits = ring + (size_t)slot * 148;
    const int u0 = start < lo ? (lo - start) / SPS : 0;
    const int u1 = (hi - start) / SPS < nsym - 1 ? (hi - start) / SPS : nsym - 1;
    for (int u = u0 + (int)threadIdx.x; u <= u1; u += 256) {
      cx av[3];                                            // a[SPS (u + 1 - q)], q = 0, 1, 2; valid: the symbol exists
      bool ok[3];
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const int k = u + 1 - q;
        ok[q] = k >= 0 && k < 148;
        const int kc = ok[q] ? k : 0;
        const float sym = (float)(2.0 * (bits[kc] & 0x01) - 1.0);
        av[q] = cmulr(T->rot[SPS * kc], sym);              // GMSKRotate, realOnly (:235-239)
      }
#pragma unroll
      for (int r = 0; r < SPS; r++) {
        const int i = start + SPS * u + r;
        cx sum = mk(0, 0);
        if (ok[0]) sum = cadd(sum, cmulr(av[0], pul[r]));              // j = r
        if (ok[1]) sum = cadd(sum, cmulr(av[1], pul[r + SPS]));        // j = r + SPS   (convolve, b real: :345-353)
        if (r == 0 && ok[2]) sum = cadd(sum, cmulr(av[2], pul[2 * SPS]));   // j = 2 SPS
        if (scale) sum = cmul(sum, mk(gv, 0.0f));                      // scaleVector(x, complex(g)) (:719-722)
        if (i >= lo && i <= hi) X[i - lo] = sum;
      }
    }
  }
}

enum { RES_IN_F32 = 0, RES_IN_I16 = 1, RES_IN_BITS = 2 };
template <int INKIND, bool OUT_I16>
__global__ __launch_bounds__(256) void k_resample(TrxResampleArgs a) {
  constexpr bool IN_I16 = INKIND == RES_IN_I16;
  // dynamic LDS, sized by the launcher to what this launch needs (a window of 1056 samples + 961 taps is 12 KB: a dozen
  // workgroups per CU instead of four): [xcap complex samples][taps]
  extern __shared__ __attribute__((aligned(16))) char res_lds[];
  cx *X = reinterpret_cast<cx *>(res_lds);
  float *TP = reinterpret_cast<float *>(res_lds + sizeof(cx) * (size_t)a.xcap);
  const int tile = blockIdx.x, w = blockIdx.y, s = blockIdx.z;
  const int o0 = a.o_skip + tile * a.OB;
  if (o0 >= a.n_out) return;
  const int o1 = (o0 + a.OB < a.n_out) ? o0 + a.OB : a.n_out;       // outputs [o0, o1)
  const int D = (a.L - 1) / 2 / a.Q;                                // :1177
  long long t0 = ((long long)(o0 + D) * a.Q) / a.P - (a.L + a.P - 1) / a.P;
  long long t1 = ((long long)(o1 - 1 + D) * a.Q) / a.P;
  const int lo = t0 < 0 ? 0 : (int)t0;
  const int hi = t1 > a.n - 1 ? a.n - 1 : (int)t1;                  // staged input samples [lo, hi] of this window
  const bool taps_lds = a.taps_lds != 0;
  // taps_lds == 2: BRANCH-MAJOR, TP[branch * pitch + k] = lpf[branch + P k] (pitch odd): an output walks its own row, so the
  // lanes of a wave -- whose branches differ by multiples of Q mod P -- stay on different banks; the linear layout walked at
  // stride P puts them on a handful of banks (96:260 and 260:768 both: 16-way conflicts)
  // Only the branches that are multiples of g = gcd(P, Q) ever occur (outputIx*Q mod P): the table holds P / g rows.
  const bool taps_bm = a.taps_lds == 2;
  const int pitch = a.tap_pitch, tg = a.tap_g;
  // staging: the loads of a round of eight go out together (a plain loop would wait for each load before issuing the next:
  // five dependent HBM round trips per workgroup was what the first version of this kernel spent its time on)
  if (taps_bm) {
    const int KT = (a.L + a.P - 1) / a.P, rows = a.P / tg;
    for (int e = threadIdx.x; e < rows * KT; e += 256) {
      const int row = e % rows, k = e / rows;
      const int fi = row * tg + a.P * k;
      TP[row * pitch + k] = fi < a.L ? a.lpf[fi] : 0.0f;
    }
  } else if (taps_lds) {
    for (int i0 = threadIdx.x; i0 < a.L; i0 += 256 * 4) {
      float tv[4];
#pragma unroll
      for (int q = 0; q < 4; q++) { const int i = i0 + 256 * q; tv[q] = i < a.L ? a.lpf[i] : 0.0f; }
#pragma unroll
      for (int q = 0; q < 4; q++) { const int i = i0 + 256 * q; if (i < a.L) TP[i] = tv[q]; }
    }
  }
  if (IN_I16) {
    const bool mix = a.mix_carriers > 0;
    const int s_raw = mix ? s / a.mix_carriers : s;         // the channeliser: carrier s % C of wideband stream s / C
    const short2 *raw = reinterpret_cast<const short2 *>(a.in) + (size_t)s_raw * a.in_stride;
    const short2 *hist = a.hist + (size_t)s_raw * a.hist_len;
    const long long base = (long long)w * a.win_step - a.hist_len;  // raw index of the window's sample 0
    const float mfreq = mix ? a.mix_freq[s - s_raw * a.mix_carriers] : 0.0f;
    for (int i0 = lo + threadIdx.x; i0 <= hi; i0 += 256 * 8) {
      short2 v[8];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int i = i0 + 256 * q;
        const long long r = base + i;
        v[q] = i <= hi ? (r < 0 ? hist[a.hist_len + r] : raw[r]) : make_short2(0, 0);
      }
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int i = i0 + 256 * q;
        cx xv = a.swap ? mk((float)v[q].y, (float)v[q].x) : mk((float)v[q].x, (float)v[q].y);   // unUSRPifyVector (:108-109)
        if (mix) {
          // frequencyShift's arithmetic (:459): z[n] = x[n] * expjLookup(phase[n]), the phase of raw sample n formed directly
          // (see TrxResampleArgs) -- every operation below is an IEEE double / float operation, the same on host and device
          const long long n = a.mix_n0 + base + (i <= hi ? i : hi);
          const double xr = (double)n * (double)mfreq;
          const double kk = floor(xr * 0.15915494309189535);
          const float phase = (float)(xr - kk * 6.283185307179586);
          xv = cmul(xv, dev_expj_lookup(a.mix_tables, phase));
        }
        if (i <= hi) X[i - lo] = xv;
      }
    }
  } else if (INKIND == RES_IN_BITS) {
    // the bursts behind this tile's samples: [start, start + len) in window coordinates, back to back
    // (a tile's span is at most TRX_RES_XCAP samples = 28 of the shortest bursts: 64 table entries from the one that
    // holds sample `lo` cover it)
    __shared__ int tb_start[64], tb_meta[64], tb_first;
    if (threadIdx.x == 0) {
      int a0 = 0, b0 = a.tx_n - 1;                         // last burst that starts at or before lo (0 if none does)
      while (a0 < b0) { const int mid = (a0 + b0 + 1) >> 1; if (a.tx_start[mid] <= lo) a0 = mid; else b0 = mid - 1; }
      tb_first = a0 < 0 ? 0 : a0;
    }
    __syncthreads();
    const int m_first = tb_first;
    const int M = a.tx_n - m_first < 64 ? a.tx_n - m_first : 64;
    if ((int)threadIdx.x < M) { tb_start[threadIdx.x] = a.tx_start[m_first + threadIdx.x]; tb_meta[threadIdx.x] = a.tx_meta[m_first + threadIdx.x]; }
    __syncthreads();
    const uint8_t *ring = reinterpret_cast<const uint8_t *>(a.in) + (size_t)s * a.in_stride * 148;
    const float *gring = a.tx_gain + (size_t)s * a.in_stride;
    // history before the first burst ever pushed (sendHistory starts as zeros) and anything no burst covers
    for (int i = lo + threadIdx.x; i <= hi; i += 256) X[i - lo] = mk(0, 0);
    __syncthreads();
    if (a.tx_sps == 4) tx_stage_tile<4>(a.tx_tables, ring, gring, tb_start, tb_meta, M, lo, hi, X);
    else if (a.tx_sps == 2) tx_stage_tile<2>(a.tx_tables, ring, gring, tb_start, tb_meta, M, lo, hi, X);
    else tx_stage_tile<1>(a.tx_tables, ring, gring, tb_start, tb_meta, M, lo, hi, X);
  } else {
    const cx *x = reinterpret_cast<const cx *>(a.in) + (size_t)s * a.in_stride + (size_t)w * a.win_step;
    for (int i0 = lo + threadIdx.x; i0 <= hi; i0 += 256 * 8) {
      cx v[8];
#pragma unroll
      for (int q = 0; q < 8; q++) { const int i = i0 + 256 * q; v[q] = i <= hi ? x[i] : mk(0, 0); }
#pragma unroll
      for (int q = 0; q < 8; q++) { const int i = i0 + 256 * q; if (i <= hi) X[i - lo] = v[q]; }
    }
  }
  __syncthreads();
  // outputIx*Q = branch + P*inOff is divided once per lane; from one of a lane's outputs to its next it grows by 256*Q
  int branch0, inOff0;
  {
    const long long oq = (long long)(o0 + (int)threadIdx.x + D) * a.Q;
    if (oq < 0x7fffffffLL) { branch0 = (int)((unsigned)oq % (unsigned)a.P); inOff0 = (int)((unsigned)oq / (unsigned)a.P); }
    else { branch0 = (int)(oq % a.P); inOff0 = (int)(oq / a.P); }
  }
  const int step_b = (int)((256LL * a.Q) % a.P), step_i = (int)((256LL * a.Q) / a.P);
  for (int o = o0 + threadIdx.x; o < o1; o += 256) {
    const int branch = branch0;                                     // (outputIx*Q) % P (:1180)
    int inOff = inOff0;                                             // (outputIx*Q - branch) / P (:1181)
    branch0 += step_b; inOff0 += step_i;
    if (branch0 >= a.P) { branch0 -= a.P; inOff0++; }
    int fi = branch;
    while (inOff >= a.n) { inOff--; fi += a.P; }                    // :1183-1186
    cx sum = mk(0, 0);
    if (taps_bm) {
      const float *row = TP + (branch / tg) * pitch + (fi - branch) / a.P;
      while (inOff >= 0 && fi < a.L) {                              // :1196-1200
        sum = cadd(sum, cmulr(X[inOff - lo], *row++));
        inOff--; fi += a.P;
      }
    } else {
      while (inOff >= 0 && fi < a.L) {                              // :1196-1200
        sum = cadd(sum, cmulr(X[inOff - lo], taps_lds ? TP[fi] : a.lpf[fi]));
        inOff--; fi += a.P;
      }
    }
    const size_t oi = (size_t)s * a.out_stride + (size_t)w * a.out_win_step + (size_t)(o - a.o_skip);
    if (OUT_I16) {
      // scaleVector(resampledVector, 13500.0): the complex (gain, 0) multiplies to (r*gain - i*0, r*0 + i*gain), which for
      // finite samples is (r*gain, i*gain) up to the sign of a zero, lost in the cast; (short) truncates toward zero
      short2 q;
      q.x = (short)(int)(sum.r * a.gain);
      q.y = (short)(int)(sum.i * a.gain);
      reinterpret_cast<short2 *>(a.out)[oi] = q;
    } else {
      reinterpret_cast<cx *>(a.out)[oi] = sum;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_rx_resample<KQ>: k_resample<int16 in> for the receive front end's shape -- one whole window per workgroup, at most 4 KQ taps
//   per output (L <= 4 KQ P: KQ = 1 for the 260 : 96 front end at four samples per symbol, KQ = 4 for the reference's own 65 : 96
//   with its 961-tap table: 15 taps per output) -- with the index walk taken out of the inner loop:
//     * the window sits in LDS between zero pads, so the reference's two edge rules (skip the taps whose sample lies past the
//       end, :1183-1186; stop at the first sample before the start, :1196) become products with zero samples -- the samples
//       are int16 values, hence finite, and adding +-0 to the running sum never changes its value;
//     * the taps are staged branch-major, TPB[branch][k] = lpf[branch + P k] (0 past the end of the filter, the reference
//       stops there), so an output's taps are ONE 16-byte LDS read instead of four reads P floats apart (the strided walk
//       cost 7 conflict cycles per LDS instruction);
//     * outputIx*Q = branch + P*inOff is divided once per lane and then advanced by 256 Q per output.
//   Same terms in the same order as k_resample: value-identical (tests/test_gpu_config4.py compares both with the oracle).
// ---------------------------------------------------------------------------------------------
template <int KQ>                                           // taps per output: at most 4 KQ (L <= 4 KQ P)
__global__ __launch_bounds__(256) void k_rx_resample(TrxResampleArgs a, int n_windows, int wpb) {
  extern __shared__ __attribute__((aligned(16))) char res_lds[];
  // [KT zero samples][n window samples][KT zero samples] -- AS RECEIVED, int16 I/Q pairs (4 bytes: the kernel is bound by its LDS
  // reads, sixteen samples per output at KQ = 4, and a wave's 4-byte reads are one pass where its 8-byte reads of converted samples
  // were two to four; the conversion is exact either way and costs two instructions per read) -- then the branch-major taps: KQ float4 per branch (one more of pitch when
  // KQ > 1: consecutive outputs sit Q mod P branches apart, and a power-of-two row pitch would put a wave's rows on a few banks).
  // A workgroup serves `wpb` consecutive windows of a stream with ONE staging of the taps; the next window's raw samples are
  // loaded (into registers) before the current window is filtered, so their latency hides under the arithmetic.
  constexpr int KT = 4 * KQ, TP = KQ > 1 ? KQ + 1 : 1, NQ = 5;      // NQ * 256 >= the longest window (the launcher checks)
  short2 *X = reinterpret_cast<short2 *>(res_lds) + KT;
  float4 *TPB = reinterpret_cast<float4 *>(res_lds + ((sizeof(short2) * (size_t)(a.n + 2 * KT) + 15) & ~(size_t)15));
  const int s = blockIdx.z;
  const int w0 = blockIdx.y * wpb, w1 = w0 + wpb < n_windows ? w0 + wpb : n_windows;
  const int D = (a.L - 1) / 2 / a.Q;                                // :1177
  const short2 *raw = reinterpret_cast<const short2 *>(a.in) + (size_t)s * a.in_stride;
  const short2 *hist = a.hist + (size_t)s * a.hist_len;
  short2 v[NQ];
  auto xf = [](short2 r) { return mk((float)r.x, (float)r.y); };
  auto fetch = [&](int w) {
    const int base = w * a.win_step - a.hist_len;                   // raw index of the window's sample 0
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int i = (int)threadIdx.x + 256 * q, r = base + i;
      v[q] = i < a.n ? (r < 0 ? hist[a.hist_len + r] : raw[r]) : make_short2(0, 0);
    }
  };
  fetch(w0);
  // The taps' rows are kept in the ORDER THE OUTPUTS VISIT THE BRANCHES: consecutive outputs (a wave's lanes) sit Q mod P branches
  // apart, so with rows in branch order a wave's 16-byte tap reads fall on rows scattered over the table -- the LDS serves sixteen
  // such reads per pass, and sixteen scattered rows share banks (half of this kernel's LDS cycles were bank conflicts:
  // SQ_LDS_BANK_CONFLICT 14.1 M of SQ_LDS_IDX_ACTIVE 28.3 M).  Row of branch br = br * (Q mod P)^-1 mod P (row_inv, from the
  // launcher; 0 when Q mod P has no inverse: rows in branch order): consecutive lanes read consecutive rows, whose odd pitch in
  // 16-byte units walks all sixteen bank groups.
  const int row_inv = a.row_inv;
  for (int e = threadIdx.x; e < a.P * KQ; e += 256) {
    const int br = e / KQ, q = e - br * KQ;
    float t[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { const int fi = br + a.P * (4 * q + k); t[k] = fi < a.L ? a.lpf[fi] : 0.0f; }
    const int row = row_inv ? (int)(((unsigned)br * (unsigned)row_inv) % (unsigned)a.P) : br;
    TPB[row * TP + q] = make_float4(t[0], t[1], t[2], t[3]);
  }
  if (threadIdx.x < 2 * KT) X[threadIdx.x < KT ? (int)threadIdx.x - KT : a.n + (int)threadIdx.x - KT] = make_short2(0, 0);
  const unsigned oq0 = (unsigned)(a.o_skip + (int)threadIdx.x + D) * (unsigned)a.Q;
  const int branch0 = (int)(oq0 % (unsigned)a.P), inOff0 = (int)(oq0 / (unsigned)a.P);
  const int step_b = (256 * a.Q) % a.P, step_i = (256 * a.Q) / a.P;
  const int row0 = row_inv ? (int)(((unsigned)branch0 * (unsigned)row_inv) % (unsigned)a.P) : branch0;
  const int step_r = row_inv ? 256 % a.P : step_b;          // a row per output: 256 outputs on
  for (int w = w0; w < w1; w++) {
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int i = (int)threadIdx.x + 256 * q;
      if (i < a.n) X[i] = a.swap ? make_short2(v[q].y, v[q].x) : v[q];   // unUSRPifyVector's order (:108-109); its int16 -> float on the way out
    }
    __syncthreads();
    if (w + 1 < w1) fetch(w + 1);
    int branch = branch0, inOff = inOff0, row = row0;
    cx *out = reinterpret_cast<cx *>(a.out) + (size_t)s * a.out_stride + (size_t)w * a.out_win_step;
    for (int o = a.o_skip + threadIdx.x; o < a.n_out; o += 256) {
      // tap k of the reference's walk (fi = branch + P k) meets sample inOff - k; samples outside [0, n) are the zero pads
      // (the launcher guarantees 0 <= inOff <= n + KT - 1, so every index lies in [-(KT - 1), n + KT - 1]); taps past the
      // filter's end are zeros -- the reference stops there, and a product with 0 adds +-0
      cx sum = mk(0, 0);
#pragma unroll
      for (int q = 0; q < KQ; q++) {
        const float4 tp = TPB[row * TP + q];
        sum = cadd(sum, cmulr(xf(X[inOff - 4 * q]), tp.x));
        sum = cadd(sum, cmulr(xf(X[inOff - 4 * q - 1]), tp.y));
        sum = cadd(sum, cmulr(xf(X[inOff - 4 * q - 2]), tp.z));
        sum = cadd(sum, cmulr(xf(X[inOff - 4 * q - 3]), tp.w));
      }
      out[o - a.o_skip] = sum;
      branch += step_b; inOff += step_i; row += step_r;
      if (branch >= a.P) { branch -= a.P; inOff++; }
      if (row >= a.P) row -= a.P;
    }
    __syncthreads();                                                // every lane is done with X before the next window lands
  }
}

// trxsig_txbe_push_bursts, fused mode: the bits (and gains) of the pushed bursts go into the per-stream burst ring
__global__ __launch_bounds__(256) void k_tx_ring_store(const uint8_t *__restrict__ bits, const float *__restrict__ gain, int S, int nb,
                                                       int head, int cap, uint8_t *__restrict__ ring, float *__restrict__ gring) {
  // a thread per 32-bit word (148 bytes = 37 words; both arrays are word-aligned: hipMalloc'd, 148 = 4 * 37), the 38th thread of a burst its gain
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= (long long)S * nb * 38) return;
  const int i = (int)(g / 38), w = (int)(g - (long long)i * 38);   // i = (stream, burst)
  const int s = i / nb, j = i - s * nb;
  const int slot = (head + j) % cap;
  if (w < 37)
    reinterpret_cast<uint32_t *>(ring + ((size_t)s * cap + slot) * 148)[w] = reinterpret_cast<const uint32_t *>(bits + (size_t)i * 148)[w];
  else
    gring[(size_t)s * cap + slot] = gain ? gain[i] : 1.0f;
}

// (the same byte by byte, for a caller's bit array that is not word-aligned)
__global__ __launch_bounds__(256) void k_tx_ring_store_bytes(const uint8_t *__restrict__ bits, const float *__restrict__ gain, int S, int nb,
                                                             int head, int cap, uint8_t *__restrict__ ring, float *__restrict__ gring) {
  const int i = blockIdx.x;                                // (stream, burst)
  const int s = i / nb, j = i - s * nb;
  const int slot = (head + j) % cap;
  if (threadIdx.x < 148) ring[((size_t)s * cap + slot) * 148 + threadIdx.x] = bits[(size_t)i * 148 + threadIdx.x];
  if (threadIdx.x == 0) gring[(size_t)s * cap + slot] = gain ? gain[i] : 1.0f;
}

// trxsig_rxfe_pop's index arrays: burst j of stream s starts at s*stride + rd + (samples of bursts 0..j-1)
__global__ __launch_bounds__(256) void k_burst_index(int S, int nb, long long stride, int rd, int tn0, int sps, int32_t *__restrict__ off,
                                                     int32_t *__restrict__ len) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= S * nb) return;
  const int s = i / nb, j = i - s * nb;
  // bursts tn0 .. tn0+j-1 precede burst j: 156 symbols each, plus one for every TN that is a multiple of 4
  const int long_before = (tn0 + j + 3) / 4 - (tn0 + 3) / 4;
  const int pos = rd + (156 * j + long_before) * sps;
  const int tn = (tn0 + j) & 7;
  off[i] = (int32_t)((long long)s * stride + pos);
  len[i] = (156 + ((tn & 3) == 0)) * sps;
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



}  // namespace

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

// outputs per workgroup: as many as keep the staged input span inside TRX_RES_XCAP (and enough workgroups in flight)
static int resample_tile(const TrxResampleArgs &a, int xcap_limit) {
  long long ob = ((long long)(xcap_limit - (a.L + a.P - 1) / a.P - 4) * a.P) / a.Q;
  if (ob > 4096) ob = 4096;
  if (ob < 64) ob = 64;                                    // (Q/P so large that 64 outputs overflow the span: not a resampler ratio)
  const int want = a.n_out - a.o_skip;
  return (int)(ob < want ? ob : want);
}

hipError_t trx_launch_resample_ex(hipStream_t st, TrxResampleArgs a, int S, int n_windows, bool in_i16, bool out_i16,
                                  TrxProfiler *prof, bool in_bits) {
  if (S <= 0 || n_windows <= 0 || a.n_out <= a.o_skip) return hipSuccess;
  // the channeliser's tiles are smaller: its staging is arithmetic (the mixer), not copying, and wants more waves in flight
  a.OB = resample_tile(a, a.mix_carriers > 0 ? 2048 : TRX_RES_XCAP);
  long long span = ((long long)(a.OB - 1) * a.Q) / a.P + (a.L + a.P - 1) / a.P + 4;      // staged samples a tile can need
  if (span > TRX_RES_XCAP) return hipErrorInvalidValue;
  if (span > a.n) span = a.n;
  if (S > 65535 || n_windows > 65535) return hipErrorInvalidValue;
  a.xcap = (int)((span + 1) & ~1LL);
  a.taps_lds = a.L <= TRX_RES_LCAP;
  size_t tap_bytes = a.taps_lds ? sizeof(float) * (size_t)a.L : 0;
  {
    const int KT = (a.L + a.P - 1) / a.P;
    const int pitch = KT | 1;
    int g = a.P, r = a.Q % a.P;
    while (r) { const int t = g % r; g = r; r = t; }        // gcd(P, Q)
    if (KT >= 3 && sizeof(float) * (size_t)(a.P / g) * pitch <= 40 * 1024) {   // branch-major taps (see k_resample)
      a.taps_lds = 2; a.tap_pitch = pitch; a.tap_g = g;
      tap_bytes = sizeof(float) * (size_t)(a.P / g) * pitch;
    }
  }
  const size_t lds = sizeof(trx_c32) * (size_t)a.xcap + tap_bytes;
  const dim3 grid((a.n_out - a.o_skip + a.OB - 1) / a.OB, n_windows, S), block(256);
  if (prof) prof->begin(TRXSIG_K_RESAMPLE, st);
  // the receive front end's shape (whole window per workgroup, <= 4 taps per output, indices inside 32 bits): k_rx_resample
  const int kq = a.L <= 4 * a.P ? 1 : (a.L <= 8 * a.P ? 2 : 4);
  const bool rx_fast = in_i16 && !out_i16 && a.OB == a.n_out - a.o_skip && a.L <= 16 * a.P && a.n <= 5 * 256 && a.P <= 1024 &&
                       (long long)(a.n_out + (a.L - 1) / 2 / a.Q + 256) * a.Q < 0x7fffffffLL && (long long)n_windows * a.win_step < 0x7fffffffLL &&
                       ((long long)(a.n_out - 1 + (a.L - 1) / 2 / a.Q) * a.Q) / a.P <= a.n + 4 * kq - 1;
  if (rx_fast) {
    const size_t lds2 = ((sizeof(short2) * (size_t)(a.n + 8 * kq) + 15) & ~(size_t)15) + sizeof(float) * 4 * (size_t)a.P * (kq > 1 ? kq + 1 : 1);
    // windows per workgroup: enough workgroups left to fill the machine several times over, else one window each
    int wpb = 1;
    while (wpb < 8 && (long long)S * (n_windows / (2 * wpb)) >= 4096) wpb *= 2;
    { const int v = trx_knob(TRX_KNOB_RXRES_WPB); if (v >= 1 && v <= 64) wpb = v; }   // (TRXSIG_TUNE_RXRES_WPB; tests: small cases through the multi-window loop)
    const dim3 g2(1, (n_windows + wpb - 1) / wpb, S);
    a.row_inv = 0;                                          // (Q mod P)^-1 mod P, if it exists: the rows of the tap table in visiting order
    for (int v = 1; v < a.P; v++)
      if (((long long)v * (a.Q % a.P)) % a.P == 1) { a.row_inv = v; break; }
    if ((long long)a.P * a.P > 0x7fffffffLL) a.row_inv = 0;
    if (trx_knob(TRX_KNOB_RXRES_ROWS) == 0) a.row_inv = 0;   // (TRXSIG_TUNE_RXRES_ROWS, A/B: rows in branch order)
    if (kq == 1) k_rx_resample<1><<<g2, block, lds2, st>>>(a, n_windows, wpb);
    else if (kq == 2) k_rx_resample<2><<<g2, block, lds2, st>>>(a, n_windows, wpb);
    else k_rx_resample<4><<<g2, block, lds2, st>>>(a, n_windows, wpb);
  } else if (in_bits && out_i16 && !in_i16) k_resample<RES_IN_BITS, true><<<grid, block, lds, st>>>(a);
  else if (in_bits) return hipErrorInvalidValue;
  else if (in_i16 && !out_i16) k_resample<RES_IN_I16, false><<<grid, block, lds, st>>>(a);
  else if (!in_i16 && out_i16) k_resample<RES_IN_F32, true><<<grid, block, lds, st>>>(a);
  else if (!in_i16 && !out_i16) k_resample<RES_IN_F32, false><<<grid, block, lds, st>>>(a);
  else return hipErrorInvalidValue;
  if (prof) prof->end(TRXSIG_K_RESAMPLE, st);
  return hipGetLastError();
}

hipError_t trx_launch_resample(hipStream_t st, const trx_c32 *in, int n, long long in_stride, int S, int P, int Q,
                               const float *lpf, int L, trx_c32 *out, long long out_stride, int nout,
                               TrxProfiler *prof) {
  TrxResampleArgs a = {};
  a.in = in; a.in_stride = in_stride; a.n = n; a.lpf = lpf; a.L = L; a.P = P; a.Q = Q; a.n_out = nout; a.out = out;
  a.out_stride = out_stride;
  return trx_launch_resample_ex(st, a, S, 1, false, false, prof);
}

hipError_t trx_launch_burst_index(hipStream_t st, int S, int nb, long long stride, int rd, int tn0, int sps, int32_t *off,
                                  int32_t *len) {
  if (S * nb <= 0) return hipSuccess;
  k_burst_index<<<dim3((S * nb + 255) / 256), dim3(256), 0, st>>>(S, nb, stride, rd, tn0, sps, off, len);
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


hipError_t trx_launch_tx_ring_store(hipStream_t st, const uint8_t *bits, const float *gain, int S, int nb, int head, int cap, uint8_t *ring,
                                    float *gring) {
  if (S * nb <= 0) return hipSuccess;
  if (((uintptr_t)bits | (uintptr_t)ring) & 3) {
    k_tx_ring_store_bytes<<<dim3(S * nb), dim3(256), 0, st>>>(bits, gain, S, nb, head, cap, ring, gring);
    return hipGetLastError();
  }
  const long long words = (long long)S * nb * 38;
  k_tx_ring_store<<<dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st>>>(bits, gain, S, nb, head, cap, ring, gring);
  return hipGetLastError();
}
