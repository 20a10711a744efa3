// trxsig_tx.hip -- transmit side and radio-format kernels: modulateBurst, polyphaseResampleVector,
// int16 / fp16 I/Q conversion.
// Numerical contract: see trxsig_dev.h / DESIGN.md (every float32 operation is the reference's, in the
// reference's order; built with -ffp-contract=off).
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

