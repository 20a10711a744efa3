// trxsig_prim.hip -- the free-standing vector primitives of sigProcLib.h as batch kernels: convolve / correlate (every span
// type, the four real/complex forms), delayVector, interpolatePoint, peakDetect, scaleVector, GMSKRotate /
// GMSKReverseRotate, decimateVector, vectorSlicer.  On the burst path they only ever run fused into the burst kernels
// (trxsig_normal.hip, trxsig_rach.hip, trxsig_demod.h); these stand-alone forms complete the sigProcLib.h surface (the
// facade, include/sigProcLib_trx.h, and config 1's sigProcLibTest call sequence go through them).  They are written for
// fidelity first: one output per lane, every sum in the reference's order with the reference's skip / break rules.
// Numerical contract: see trxsig_dev.h / DESIGN.md (every float32 operation is the reference's, in the reference's order;
// built with -ffp-contract=off).
#include "trxsig_dev.h"

namespace {

// ---------------------------------------------------------------------------------------------
// convolve (sigProcLib.cpp:267-408, symmetry NONE) and correlate (:474-503).  Output t (index startIndex + t into the
// full convolution) = sum over j = 0 .. Lb-1 of a[t-j] b[j], leaving the loop at the first t-j < 0 and skipping t-j >= La
// (:322-366).  CORR: b is read reversed and conjugated (:480-498), i.e. b'[j] = conj(b[Lb-1-j]).
// flags: bit 0 = a real-only, bit 1 = b real-only (the four arithmetic forms of :326-365).
// ---------------------------------------------------------------------------------------------
template <bool CORR>
__global__ __launch_bounds__(256) void k_convolve(const cx *__restrict__ a, const int32_t *__restrict__ a_off,
                                                  const int32_t *__restrict__ a_len, const cx *__restrict__ b, int Lb, int span,
                                                  int flags, int cust_start, int cust_len, cx *__restrict__ out,
                                                  const int32_t *__restrict__ out_off) {
  const int v = blockIdx.y;
  const int La = a_len[v];
  int start, osz;
  switch (span) {
    case TRXSIG_FULL_SPAN: start = 0; osz = La + Lb - 1; break;
    case TRXSIG_OVERLAP_ONLY: start = La; osz = (La > Lb ? La - Lb : Lb - La) + 1; break;
    case TRXSIG_START_ONLY: start = 0; osz = La; break;
    case TRXSIG_WITH_TAIL: start = Lb; osz = La; break;
    case TRXSIG_NO_DELAY: start = (Lb & 1) ? Lb / 2 : Lb / 2 - 1; osz = La; break;
    default: start = cust_start; osz = cust_len; break;    // CUSTOM (Transceiver52M/sigProcLib.cpp:301-304)
  }
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (La <= 0 || o >= osz) return;
  const cx *av = a + a_off[v];
  const int t = start + o;
  const bool ar = flags & 1, br = flags & 2;
  cx sum = mk(0, 0);
  float fsum = 0.0f;
  if (!CORR && (flags & 4)) {
    // b->getSymmetry() == ABSSYM (:369-398): half the taps, each applied to a[t-j] and to its partner a[t-Lb+j] (one
    // sample further than the mirror image, as the reference has it); complex arithmetic whatever the real-only marks
    // say.  The reference reads the partner without an upper bound in its fourth arm: a partner at or beyond a's end counts
    // as zero here.
    const int half = (Lb & 1) ? (Lb + 1) / 2 : Lb / 2;
    for (int j = 0; j < half; j++) {
      const int ia = t - j, is = t - Lb + j;
      if (ia < 0) break;
      const cx sv = (is >= 0 && is < La) ? av[is] : mk(0, 0);
      if (ia == is) sum = cadd(sum, cmul(av[ia], b[j]));
      else if (ia < La && is >= 0) sum = cadd(sum, cmul(cadd(av[ia], sv), b[j]));
      else if (ia < La) sum = cadd(sum, cmul(av[ia], b[j]));
      else if (is >= 0) sum = cadd(sum, cmul(sv, b[j]));
    }
    out[out_off[v] + o] = sum;
    return;
  }
  for (int j = 0; j < Lb; j++) {
    const int ia = t - j;
    if (ia < 0) break;                                     // :327 "if (aP < aStart) break"
    if (ia >= La) continue;
    cx bj = CORR ? b[Lb - 1 - j] : b[j];
    if (CORR) bj = br ? mk(bj.r, 0.0f) : mk(bj.r, -bj.i);  // :485-496
    const cx aj = av[ia];
    if (ar && br) fsum += aj.r * bj.r;                     // :326-335
    else if (ar) sum = cadd(sum, cmulr(bj, aj.r));         // :336-345  (*bP)*(aP->real())
    else if (br) sum = cadd(sum, cmulr(aj, bj.r));         // :346-355  (*aP)*(bP->real())
    else sum = cadd(sum, cmul(aj, bj));                    // :356-365  (*aP)*(*bP)
  }
  out[out_off[v] + o] = (ar && br) ? mk(fsum, 0.0f) : sum;
}

// ---------------------------------------------------------------------------------------------
// delayVector (:573-616): fractional part by the 21-tap table sinc convolved NO_DELAY when |frac| > 1e-2, then the
// integer shift with zero fill.  One workgroup per vector; out must not alias in.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_delay_vector(const TrxTables *__restrict__ T, const cx *__restrict__ in,
                                                      const int32_t *__restrict__ off, const int32_t *__restrict__ len,
                                                      const float *__restrict__ delay, int real_only, cx *__restrict__ out) {
  __shared__ float tap[21];
  const int v = blockIdx.x;
  const int n = len[v];
  const cx *x = in + off[v];
  cx *y = out + off[v];
  const float d = delay[v];
  if (!(fabsf(d) <= TRXSIG_MAX_INDEX)) {                   // beyond what the reference's sinc can reduce (or inf / NaN): zeros (trxsig.h)
    for (int k = threadIdx.x; k < n; k += 256) y[k] = mk(0, 0);
    return;
  }
  const int io = (int)floorf(d);                           // :577
  const float frac = d - (float)io;                        // :578
  const bool filt = fabs((double)frac) > 1e-2;             // :582
  if (threadIdx.x < 21) tap[threadIdx.x] = dev_sinc(T->sinT, TRX_PI_F * ((float)((int)threadIdx.x - 10) - frac));   // :588
  __syncthreads();
  for (int k = threadIdx.x; k < n; k += 256) {
    // wBurst[k] = shifted[k - io] where that exists (:597-613), else 0
    const int t = k - io;
    cx r = mk(0, 0);
    if (t >= 0 && t < n) {
      if (filt) {
        for (int j = 0; j < 21; j++) {                     // convolve(&wBurst, sincVector, NO_DELAY): start = 10, b real-only
          const int ia = t + 10 - j;
          if (ia < 0) break;
          if (ia < n) r = real_only ? mk(r.r + x[ia].r * tap[j], 0.0f) : cadd(r, cmulr(x[ia], tap[j]));   // :326-335 / :346-355
        }
      } else {
        r = x[t];
      }
    }
    y[k] = r;
  }
}

// interpolatePoint (:639-659) by the lanes of one wave: lane q forms term i = start + q, the sum runs i ascending on
// every lane (readlane), so the value is the reference's.  More than 64 terms (only when the clamp of :645-646 opens the
// range to the whole vector) are handled in rounds.
__device__ __forceinline__ cx wave_interpolate(const TrxTables *__restrict__ T, const cx *__restrict__ x, int n, float ix,
                                               bool real_only, int lane) {
  int start = (int)(floorf(ix) - 10);                      // :643
  if (start < 0) start = 0;
  int end = (int)(floorf(ix) + 11);                        // :645
  if ((unsigned)end > (unsigned)(n - 1)) end = n - 1;      // :646 (unsigned compare: a negative end opens the range)
  cx p = mk(0, 0);
  for (int base = start; base < end; base += 64) {
    const int i = base + lane;
    cx term = mk(0, 0);
    if (i < end) {
      const float s = dev_sinc(T->sinT, TRX_PI_F * ((float)i - ix));   // :651
      term = real_only ? mk(x[i].r * s, 0.0f) : cmulr(x[i], s);        // :651 / :655
    }
    const int cnt = end - base < 64 ? end - base : 64;
    for (int q = 0; q < cnt; q++)
      p = cadd(p, mk(__shfl(term.r, q, 64), __shfl(term.i, q, 64)));
  }
  return p;
}

__global__ __launch_bounds__(64) void k_interpolate_point(const TrxTables *__restrict__ T, const cx *__restrict__ in,
                                                          const int32_t *__restrict__ off, const int32_t *__restrict__ len,
                                                          const float *__restrict__ ix, int real_only, cx *__restrict__ out) {
  const int v = blockIdx.x;
  const int n = len[v];
  if (n <= 0 || !(fabsf(ix[v]) <= TRXSIG_MAX_INDEX)) { if (threadIdx.x == 0) out[v] = mk(0, 0); return; }   // (trxsig.h: TRXSIG_MAX_INDEX)
  const cx p = wave_interpolate(T, in + off[v], n, ix[v], real_only != 0, threadIdx.x);
  if (threadIdx.x == 0) out[v] = p;
}

// ---------------------------------------------------------------------------------------------
// peakDetect (:663-711): first maximum of |x|^2 (strict >), the power sum in index order, the early/late bisection with
// interpolatePoint, the interpolated peak, avgPwr = (sum - |peak|^2) / (n - 1).  One wave per vector.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_peak_detect(const TrxTables *__restrict__ T, const cx *__restrict__ in,
                                                    const int32_t *__restrict__ off, const int32_t *__restrict__ len,
                                                    cx *__restrict__ peak_out, float *__restrict__ index_out,
                                                    float *__restrict__ avgpwr_out) {
  const int v = blockIdx.x, lane = threadIdx.x;
  const int n = len[v];
  const cx *x = in + off[v];
  if (n <= 0) { if (lane == 0) { peak_out[v] = mk(0, 0); if (index_out) index_out[v] = 0; if (avgpwr_out) avgpwr_out[v] = 0; } return; }
  float bestP = 0.0f, sumPower = 0.0f;
  int bestT = -1;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    const float p = i < n ? norm2(x[i]) : 0.0f;
    if (i < n && p > bestP) { bestP = p; bestT = i; }      // per lane: indices ascending, strict > keeps the first
    const int cnt = n - base < 64 ? n - base : 64;
    for (int q = 0; q < cnt; q++) sumPower += __shfl(p, q, 64);   // :679, index order
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) {
    const float oP = __shfl_xor(bestP, m, 64);
    const int oT = __shfl_xor(bestT, m, 64);
    const bool take = (oP > bestP) || (oP == bestP && oT >= 0 && (bestT < 0 || oT < bestT));
    if (take) { bestP = oP; bestT = oT; }
  }
  float maxIndex = (float)bestT;                           // -1 when every sample is zero (:669)
  float early = maxIndex - 1, late = maxIndex + 1;         // :684-685
  float incr = 0.5f;
  while (incr > 1.0f / 1024.0f) {                          // :688
    const cx e = wave_interpolate(T, x, n, early, false, lane);
    const cx l = wave_interpolate(T, x, n, late, false, lane);
    const float ne = norm2(e), nl = norm2(l);              // Complex < and > compare norm2 (Complex.h:109-110)
    if (ne < nl) early += incr;
    else if (ne > nl) early -= incr;
    else break;
    incr = incr / 2.0f;
    late = early + 2.0f;
  }
  maxIndex = early + 1.0f;
  const cx pk = wave_interpolate(T, x, n, maxIndex, false, lane);
  if (lane == 0) {
    peak_out[v] = pk;
    if (index_out) index_out[v] = maxIndex;
    if (avgpwr_out) avgpwr_out[v] = (sumPower - norm2(pk)) / (float)(unsigned)(n - 1);   // :707 (size()-1 is unsigned)
  }
}

// ---------------------------------------------------------------------------------------------
// energyDetect (:916-932; the 52 MHz variant steps four samples at a time, Transceiver52M/sigProcLib.cpp:946-963): the
// window's powers summed in index order, avgPwr = energy / windowLength, the decision avgPwr > threshold^2.  One wave
// per vector.  (With step 4 the reference can read past a short vector; such samples count as zero here.)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_energy_detect(const cx *__restrict__ in, const int32_t *__restrict__ off,
                                                      const int32_t *__restrict__ len, unsigned window, int step, float thresh,
                                                      float *__restrict__ avgpwr_out, uint8_t *__restrict__ ok_out) {
  const int v = blockIdx.x, lane = threadIdx.x;
  const int n = len[v];
  const cx *x = in + off[v];
  unsigned w = window;
  if (w > (unsigned)(n < 0 ? 0 : n)) w = (unsigned)(n < 0 ? 0 : n);   // :924
  float energy = 0.0f;
  for (unsigned base = 0; base < w; base += 64) {
    const unsigned i = base + lane;
    const long long ix = (long long)i * step;
    const float p = (i < w && ix < n) ? norm2(x[ix]) : 0.0f;
    const int cnt = w - base < 64 ? (int)(w - base) : 64;
    for (int q = 0; q < cnt; q++) energy += __shfl(p, q, 64);       // :925-928, index order
  }
  if (lane == 0) {
    const float avg = energy / (float)w;                           // :929 (0/0 for an empty vector, as the reference)
    if (avgpwr_out) avgpwr_out[v] = avg;
    if (ok_out) ok_out[v] = avg > thresh * thresh;                 // :931
  }
}

// ---------------------------------------------------------------------------------------------
// element-wise: scaleVector (:713-730), GMSKRotate / GMSKReverseRotate (:232-264), vectorSlicer (:507-519),
// decimateVector (:1039-1053)
// ---------------------------------------------------------------------------------------------
enum { EW_SCALE = 0, EW_ROTATE = 1, EW_REVROT = 2, EW_SLICE = 3, EW_OFFSET = 4 };
template <int OP>
__global__ __launch_bounds__(256) void k_elementwise(const TrxTables *__restrict__ T, cx *__restrict__ x, const int32_t *__restrict__ off,
                                                     const int32_t *__restrict__ len, const cx *__restrict__ scale, int real_only) {
  const int v = blockIdx.y;
  const int n = len[v];
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  cx *p = x + off[v] + k;
  const cx xv = *p;
  cx r;
  if (OP == EW_SCALE) {
    const cx s = scale[v];
    r = real_only ? cmulr(s, xv.r) : cmul(xv, s);          // :725 xP->real()*scale -> Complex*Real; :719 *xP * scale
  } else if (OP == EW_OFFSET) {
    const cx o = scale[v];                                 // offsetVector (:760-777): *xP += offset / xP->real() + offset
    r = real_only ? mk(o.r + xv.r, o.i) : cadd(xv, o);
  } else if (OP == EW_ROTATE || OP == EW_REVROT) {
    if (k >= 157 * (int)T->sps) return;                    // the tables hold 157*sps entries (:215-216)
    const cx rt = OP == EW_ROTATE ? T->rot[k] : T->rev[k];
    r = real_only ? cmulr(rt, xv.r) : cmul(rt, xv);        // :237, :243  *rotPtr * (*xPtr)
  } else {
    // (complex)(0.5*(re+1.0F)) in double, then the clamps (:513-515)
    float sv = (float)(0.5 * (double)(xv.r + 1.0F));
    if (sv > 1.0f) sv = 1.0f;
    if (sv < 0.0f) sv = 0.0f;
    r = mk(sv, 0.0f);
  }
  *p = r;
}

// ---------------------------------------------------------------------------------------------
// The part of sigProcLib.h the burst path never calls (sigProcLib.h:108-111, 149-153, 184-185, 352-354).
// vectorNorm2 / vectorPower (:146-160): the powers summed in index order.  One wave per vector.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_vector_norm2(const cx *__restrict__ in, const int32_t *__restrict__ off,
                                                     const int32_t *__restrict__ len, float *__restrict__ norm2_out,
                                                     float *__restrict__ power_out) {
  const int v = blockIdx.x, lane = threadIdx.x;
  const int n = len[v] < 0 ? 0 : len[v];
  const cx *x = in + off[v];
  float energy = 0.0f;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    const float p = i < n ? norm2(x[i]) : 0.0f;
    const int cnt = n - base < 64 ? n - base : 64;
    for (int q = 0; q < cnt; q++) energy += __shfl(p, q, 64);
  }
  if (lane == 0) {
    if (norm2_out) norm2_out[v] = energy;
    if (power_out) power_out[v] = energy / (float)(size_t)n;          // vectorNorm2(x)/x.size()
  }
}

// frequencyShift (:432-471): y[k] = x[k] * expjLookup(phase_k), phase_k = startPhase + freq + ... + freq (k sequential float
// additions).  One wave per vector; every lane runs the whole chain of additions and keeps the phases of its own elements.
__global__ __launch_bounds__(64) void k_frequency_shift(const TrxTables *__restrict__ T, const cx *__restrict__ in,
                                                        const int32_t *__restrict__ off, const int32_t *__restrict__ len,
                                                        const float *__restrict__ freq, const float *__restrict__ start,
                                                        int real_only, cx *__restrict__ out, float *__restrict__ final_phase) {
  const int v = blockIdx.x, lane = threadIdx.x;
  const int n = len[v] < 0 ? 0 : len[v];
  const cx *x = in + off[v];
  cx *y = out + off[v];
  const float f = freq[v];
  float phase = start[v];
  for (int base = 0; base < n; base += 64) {
    float mine = phase;
    const int cnt = n - base < 64 ? n - base : 64;
    for (int q = 0; q < cnt; q++) {
      mine = q == lane ? phase : mine;
      phase += f;                                          // :454, :460
    }
    const int i = base + lane;
    if (i < n) {
      const cx e = dev_expj_lookup(T, mine);
      const cx xv = x[i];
      y[i] = real_only ? cmulr(e, xv.r) : cmul(xv, e);     // :453 expjLookup(phase)*real(); :459 (*xP)*expjLookup(phase)
    }
  }
  if (lane == 0 && final_phase) final_phase[v] = phase;
}

// addVector (:746-758): x[k] = x[k] + y[k] over the shorter of the two
__global__ __launch_bounds__(256) void k_add_vector(cx *__restrict__ x, const int32_t *__restrict__ xoff, const int32_t *__restrict__ xlen,
                                                    const cx *__restrict__ y, const int32_t *__restrict__ yoff,
                                                    const int32_t *__restrict__ ylen) {
  const int v = blockIdx.y;
  const int n = xlen[v] < ylen[v] ? xlen[v] : ylen[v];
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  cx *p = x + xoff[v] + k;
  *p = cadd(*p, y[yoff[v] + k]);
}

// resampleVector (:1213-1243) AS IT BEHAVES: the loop never advances its output iterator, so every interpolated value is
// written to element 0 and the rest of the freshly allocated (zeroed) vector stays zero.  A lane per vector.
__global__ __launch_bounds__(64) void k_resample_linear(const cx *__restrict__ in, const int32_t *__restrict__ off,
                                                        const int32_t *__restrict__ len, int B, float expFactor,
                                                        const cx *__restrict__ end_point, cx *__restrict__ out,
                                                        const int32_t *__restrict__ out_off) {
  const int v = blockIdx.x * 64 + threadIdx.x;
  if (v >= B) return;
  const int n = len[v] < 0 ? 0 : len[v];
  const cx *x = in + off[v];
  cx *y = out + out_off[v];
  const int nout = (int)ceilf((float)(size_t)n * expFactor);
  for (int k = 0; k < nout; k++) y[k] = mk(0, 0);
  const cx endPoint = end_point[v];
  float t = 0.0f;
  while (nout > 0) {
    const unsigned tLow = (unsigned)floorf(t);
    const unsigned tHigh = tLow + 1;
    if (tLow > (unsigned)n - 1) break;
    if (tHigh > (unsigned)n) break;
    const cx lowPoint = x[tLow];
    const cx highPoint = (tHigh == (unsigned)n) ? endPoint : x[tHigh];
    const cx a = mk((float)tHigh - t, 0.0f);               // complex a = (tHigh-t)
    const cx b = mk(t - (float)tLow, 0.0f);
    y[0] = cadd(cmul(a, lowPoint), cmul(b, highPoint));
    t = (float)((double)t + 1.0 / (double)expFactor);      // t += 1.0/expFactor
  }
}

__global__ __launch_bounds__(256) void k_decimate(const cx *__restrict__ in, const int32_t *__restrict__ off,
                                                  const int32_t *__restrict__ len, int factor, cx *__restrict__ out,
                                                  const int32_t *__restrict__ out_off) {
  const int v = blockIdx.y;
  const int n = len[v];
  const int k = blockIdx.x * 256 + threadIdx.x;
  // the reference writes one element per i = 0, factor, 2 factor, .. < n into a vector of n / factor elements (:1045-1050):
  // n must be a multiple of factor (else it writes past its allocation); only the n / factor elements are produced here
  if (k >= n / factor) return;
  out[out_off[v] + k] = in[off[v] + (size_t)k * factor];
}

}  // namespace

int trx_convolve_out_len(int La, int Lb, int span, int cust_len) {
  switch (span) {
    case TRXSIG_FULL_SPAN: return La + Lb - 1;
    case TRXSIG_OVERLAP_ONLY: return (La > Lb ? La - Lb : Lb - La) + 1;
    case TRXSIG_START_ONLY: case TRXSIG_WITH_TAIL: case TRXSIG_NO_DELAY: return La;
    case TRXSIG_CUSTOM: return cust_len;
  }
  return -1;
}

hipError_t trx_launch_convolve(hipStream_t st, const trx_c32 *a, const int32_t *a_off, const int32_t *a_len, int B, int max_out,
                               const trx_c32 *b, int Lb, int span, int flags, int correlate, int cust_start, int cust_len,
                               trx_c32 *out, const int32_t *out_off) {
  if (B <= 0 || max_out <= 0) return hipSuccess;
  const dim3 grid((max_out + 255) / 256, B), block(256);
  if (correlate) k_convolve<true><<<grid, block, 0, st>>>(a, a_off, a_len, b, Lb, span, flags, cust_start, cust_len, out, out_off);
  else k_convolve<false><<<grid, block, 0, st>>>(a, a_off, a_len, b, Lb, span, flags, cust_start, cust_len, out, out_off);
  return hipGetLastError();
}

hipError_t trx_launch_delay_vector(hipStream_t st, const TrxTables *dT, const trx_c32 *in, const int32_t *off, const int32_t *len,
                                   int B, const float *delay, int real_only, trx_c32 *out) {
  if (B <= 0) return hipSuccess;
  k_delay_vector<<<dim3(B), dim3(256), 0, st>>>(dT, in, off, len, delay, real_only, out);
  return hipGetLastError();
}

hipError_t trx_launch_interpolate_point(hipStream_t st, const TrxTables *dT, const trx_c32 *in, const int32_t *off,
                                        const int32_t *len, int B, const float *ix, int real_only, trx_c32 *out) {
  if (B <= 0) return hipSuccess;
  k_interpolate_point<<<dim3(B), dim3(64), 0, st>>>(dT, in, off, len, ix, real_only, out);
  return hipGetLastError();
}

hipError_t trx_launch_peak_detect(hipStream_t st, const TrxTables *dT, const trx_c32 *in, const int32_t *off, const int32_t *len,
                                  int B, trx_c32 *peak, float *index, float *avgpwr) {
  if (B <= 0) return hipSuccess;
  k_peak_detect<<<dim3(B), dim3(64), 0, st>>>(dT, in, off, len, peak, index, avgpwr);
  return hipGetLastError();
}

hipError_t trx_launch_energy_detect(hipStream_t st, const trx_c32 *in, const int32_t *off, const int32_t *len, int B, unsigned window,
                                    int step, float thresh, float *avgpwr, uint8_t *ok) {
  if (B <= 0) return hipSuccess;
  k_energy_detect<<<dim3(B), dim3(64), 0, st>>>(in, off, len, window, step, thresh, avgpwr, ok);
  return hipGetLastError();
}

hipError_t trx_launch_elementwise(hipStream_t st, int op, const TrxTables *dT, trx_c32 *x, const int32_t *off, const int32_t *len,
                                  int B, int max_len, const trx_c32 *scale, int real_only) {
  if (B <= 0 || max_len <= 0) return hipSuccess;
  const dim3 grid((max_len + 255) / 256, B), block(256);
  switch (op) {
    case EW_SCALE: k_elementwise<EW_SCALE><<<grid, block, 0, st>>>(dT, x, off, len, scale, real_only); break;
    case EW_ROTATE: k_elementwise<EW_ROTATE><<<grid, block, 0, st>>>(dT, x, off, len, scale, real_only); break;
    case EW_REVROT: k_elementwise<EW_REVROT><<<grid, block, 0, st>>>(dT, x, off, len, scale, real_only); break;
    case EW_SLICE: k_elementwise<EW_SLICE><<<grid, block, 0, st>>>(dT, x, off, len, scale, real_only); break;
    case EW_OFFSET: k_elementwise<EW_OFFSET><<<grid, block, 0, st>>>(dT, x, off, len, scale, real_only); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t trx_launch_decimate(hipStream_t st, const trx_c32 *in, const int32_t *off, const int32_t *len, int B, int max_len,
                               int factor, trx_c32 *out, const int32_t *out_off) {
  if (B <= 0 || max_len <= 0) return hipSuccess;
  k_decimate<<<dim3((max_len / factor + 255) / 256 + 1, B), dim3(256), 0, st>>>(in, off, len, factor, out, out_off);
  return hipGetLastError();
}

hipError_t trx_launch_vector_norm2(hipStream_t st, const trx_c32 *in, const int32_t *off, const int32_t *len, int B, float *norm2_out,
                                   float *power_out) {
  if (B <= 0) return hipSuccess;
  k_vector_norm2<<<dim3(B), dim3(64), 0, st>>>(in, off, len, norm2_out, power_out);
  return hipGetLastError();
}
float trx_frequency_shift_max_phase(void) { return TRX_FSHIFT_MAXPHASE; }
hipError_t trx_launch_frequency_shift(hipStream_t st, const TrxTables *dT, const trx_c32 *in, const int32_t *off, const int32_t *len,
                                      int B, const float *freq, const float *start, int real_only, trx_c32 *out, float *final_phase) {
  if (B <= 0) return hipSuccess;
  k_frequency_shift<<<dim3(B), dim3(64), 0, st>>>(dT, in, off, len, freq, start, real_only, out, final_phase);
  return hipGetLastError();
}
hipError_t trx_launch_add_vector(hipStream_t st, trx_c32 *x, const int32_t *xoff, const int32_t *xlen, const trx_c32 *y,
                                 const int32_t *yoff, const int32_t *ylen, int B, int max_len) {
  if (B <= 0 || max_len <= 0) return hipSuccess;
  k_add_vector<<<dim3((max_len + 255) / 256, B), dim3(256), 0, st>>>(x, xoff, xlen, y, yoff, ylen);
  return hipGetLastError();
}
hipError_t trx_launch_resample_linear(hipStream_t st, const trx_c32 *in, const int32_t *off, const int32_t *len, int B, float exp_factor,
                                      const trx_c32 *end_point, trx_c32 *out, const int32_t *out_off) {
  if (B <= 0) return hipSuccess;
  k_resample_linear<<<dim3((B + 63) / 64), dim3(64), 0, st>>>(in, off, len, B, exp_factor, end_point, out, out_off);
  return hipGetLastError();
}
