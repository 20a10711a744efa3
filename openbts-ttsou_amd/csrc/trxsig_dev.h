// trxsig_dev.h -- device-side helpers shared by every kernel file: Complex<float> arithmetic exactly as the
// reference's Complex.h does it, the wave-local LDS fence, the reference's table sin/sinc, midamble tap
// classes (exact-product FMA form), tuning macros.
//
// Numerical contract: every float32 operation is the reference's operation, in the reference's
// order, separately rounded.  All device code is compiled with -ffp-contract=off (no v_fma/v_mac is ever
// formed from a*b+c) and with hipcc's default correctly-rounded division and square root, so the outputs are
// bit-identical to Transceiver/sigProcLib.cpp built for x86-64.  Where a sum's order is changed for
// parallelism the comment says why the result cannot change (only additions of +-0 are skipped or reordered).
#pragma once
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

// Whole-wave reductions without the LDS crossbar: four DPP steps inside each row of 16 lanes (lane ^ 1, lane ^ 2, the
// row's half mirror, the row's mirror: every lane then holds its row's result), the four row results through readlane and
// three more combines.  A __shfl_xor butterfly is six ds_bpermute round trips (~100+ cycles each on a wave that has little
// company); this is ~0.1 k cycles.  The order of a sum differs from the butterfly's: only for values whose order is free.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ float wave_sum_any_order(float v) {          // the total, in every lane
  v += dpp_f<0xB1>(v);                                      // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);                                      // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);                                     // row_half_mirror
  v += dpp_f<0x140>(v);                                     // row_mirror
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return ((r0 + r1) + r2) + r3;
}
// argmax of (P, T) over the wave with the rule "larger P wins; equal P: the smaller valid T (T >= 0) wins"; result in every lane
__device__ __forceinline__ void argmax_take(float &P, int &T, float oP, int oT) {
  const bool take = (oP > P) || (oP == P && oT >= 0 && (T < 0 || oT < T));
  P = take ? oP : P;
  T = take ? oT : T;
}
__device__ __forceinline__ void wave_argmax(float &P, int &T) {
  argmax_take(P, T, dpp_f<0xB1>(P), dpp_i<0xB1>(T));
  argmax_take(P, T, dpp_f<0x4E>(P), dpp_i<0x4E>(T));
  argmax_take(P, T, dpp_f<0x141>(P), dpp_i<0x141>(T));
  argmax_take(P, T, dpp_f<0x140>(P), dpp_i<0x140>(T));
  float bp = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(P), 0));
  int bt = __builtin_amdgcn_readlane(T, 0);
#pragma unroll
  for (int r = 16; r < 64; r += 16)
    argmax_take(bp, bt, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(P), r)), __builtin_amdgcn_readlane(T, r));
  P = bp; T = bt;
}

// Packed float32 pairs (v_pk_mul_f32 / v_pk_add_f32: both halves are separate IEEE operations, nothing is fused).  Worth it
// only where ONE wave's instruction count is the limit (the decision-feedback recursion: a single wave per 64 bursts issues
// an instruction every ~5 cycles whatever it is); throughput-bound kernels gain nothing (4.3 cycles per packed instruction).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk(cx z) { v2f v = {z.r, z.i}; return v; }
// Complex<float>::operator* (Complex.h:83): (x.r*a.r - x.i*a.i, x.r*a.i + x.i*a.r), every product and sum rounded separately
__device__ __forceinline__ v2f pk_cmul(v2f x, v2f a) {
  v2f p, q, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(p) : "v"(x), "v"(a));                  // (x.r*a.r, x.r*a.i)
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(q) : "v"(x), "v"(a));     // (x.i*a.i, x.i*a.r)
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(p), "v"(q));                     // (p.lo - q.lo, p.hi + q.hi)
  return r;
}
__device__ __forceinline__ v2f pk_mul(v2f x, v2f a) {       // (x.lo*a.lo, x.hi*a.hi): Complex * Real with a = (t, t) (Complex.h:84)
  v2f r;
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(a));
  return r;
}
// Complex * Real (Complex.h:84) with the real factor taken from one half of a register pair that holds two consecutive taps:
// HI = 0: (x.r*t.lo, x.i*t.lo), HI = 1: (x.r*t.hi, x.i*t.hi) -- a row of real taps feeds packed multiplies without being duplicated
template <int HI>
__device__ __forceinline__ v2f pk_mul_tap(v2f x, v2f t) {
  v2f r;
  if (HI) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(x), "v"(t));
  else asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(t));
  return r;
}
__device__ __forceinline__ v2f pk_csub(v2f x, v2f a) {
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(a));
  return r;
}
__device__ __forceinline__ cx unpk(v2f v) { return mk(v.x, v.y); }
__device__ __forceinline__ v2f pk_cadd(v2f x, v2f a) {
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(a));
  return r;
}

// How a burst's samples are stored in HBM: complex float32 (the reference's Complex<float>, 8 bytes) or fp16 I/Q pairs
// (4 bytes; BASELINE config 5).  Widening fp16 -> float32 is exact, so every kernel computes in float32 on the same
// values either way.  ld(): sample i; ld2(): samples 2q, 2q+1 as one float4 (needs the base 16 / 8 byte aligned).
struct SmpC32 {
  static constexpr int kBytes = 8;
  static __device__ __forceinline__ cx ld(const void *base, long long i) { return reinterpret_cast<const cx *>(base)[i]; }
  typedef cx raw_t;                                        // ldraw(): the stored sample as is (many loads in flight); widen(): to float32
  static __device__ __forceinline__ raw_t ldraw(const void *base, long long i) { return reinterpret_cast<const cx *>(base)[i]; }
  static __device__ __forceinline__ raw_t zero() { return mk(0, 0); }
  static __device__ __forceinline__ cx widen(raw_t r) { return r; }
  static __device__ __forceinline__ float4 ld2(const void *base, long long first, int q) {
    return reinterpret_cast<const float4 *>(reinterpret_cast<const cx *>(base) + first)[q];
  }
};
struct SmpF16 {
  static constexpr int kBytes = 4;
  static __device__ __forceinline__ cx ld(const void *base, long long i) {
    const float2 v = __half22float2(reinterpret_cast<const __half2 *>(base)[i]);
    return mk(v.x, v.y);
  }
  typedef unsigned raw_t;
  static __device__ __forceinline__ raw_t ldraw(const void *base, long long i) { return reinterpret_cast<const unsigned *>(base)[i]; }
  static __device__ __forceinline__ raw_t zero() { return 0u; }
  static __device__ __forceinline__ cx widen(raw_t r) {
    const float2 v = __half22float2(*reinterpret_cast<const __half2 *>(&r));
    return mk(v.x, v.y);
  }
  static __device__ __forceinline__ float4 ld2(const void *base, long long first, int q) {
    const uint2 u = reinterpret_cast<const uint2 *>(reinterpret_cast<const __half2 *>(base) + first)[q];
    const float2 a = __half22float2(*reinterpret_cast<const __half2 *>(&u.x)), b = __half22float2(*reinterpret_cast<const __half2 *>(&u.y));
    return make_float4(a.x, a.y, b.x, b.y);
  }
};

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
// A fused multiply-add in a STEERING pass: an approximate correlation that only decides which lags get recomputed with
// the reference's exact arithmetic (k_rach_*, the steered midamble correlator); nothing computed with it is ever
// handed on.  Marked so that tools/asm_stats.py and tests/test_no_fma_contraction.py can tell it from a contraction.
__device__ __forceinline__ float fma_steer(float a, float b, float c) {
  float r;
  asm("v_fma_f32 %0, %1, %2, %3 ; steering" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float fma_steer_s(float a, float b_sgpr, float c) {
  float r;
  asm("v_fma_f32 %0, %1, %2, %3 ; steering" : "=v"(r) : "v"(a), "s"(b_sgpr), "v"(c));
  return r;
}
// x * a as Complex<float>::operator* computes it (Complex.h:83), a in SGPRs, CLS = the tap's class
__device__ __forceinline__ cx cmul_tap(cx x, cx a, int cls) {
  // (one asm statement for both components: hipcc puts an s_nop behind every inline-asm block)
  if (cls == 1) {                                          // a.r = +-1: x.r*a.r and x.i*a.r are exact
    const float p = x.i * a.i, q = x.r * a.i;
    cx z;
    asm("v_fma_f32 %0, %2, %4, -%5 ; exact-product\n\tv_fma_f32 %1, %3, %4, %6 ; exact-product"
        : "=&v"(z.r), "=v"(z.i) : "v"(x.r), "v"(x.i), "s"(a.r), "v"(p), "v"(q));
    return z;                                              // (x.r*a.r - p, x.i*a.r + q)
  }
  if (cls == 2) {                                          // a.i = +-1: x.i*a.i and x.r*a.i are exact
    const float p = x.r * a.r, q = x.i * a.r;
    cx z;
    asm("v_fma_f32 %0, %3, -%4, %5 ; exact-product\n\tv_fma_f32 %1, %2, %4, %6 ; exact-product"
        : "=&v"(z.r), "=v"(z.i) : "v"(x.r), "v"(x.i), "s"(a.i), "v"(p), "v"(q));
    return z;                                              // (p - x.i*a.i, x.r*a.i + q)
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

// A workgroup barrier for data that travels through LDS only.  __syncthreads() is a barrier AND a fence over every address space:
// the compiler puts s_waitcnt vmcnt(0) in front of it, so a wave that has just issued global stores (results on their way out) sits
// there until they have landed -- microseconds, at every step of a pipelined kernel.  Here only the LDS traffic is waited for.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ void wave_lds_fence() {
  // LDS operations of one wave execute in issue order; this only stops the compiler from
  // reordering them across the point where lanes start reading what other lanes wrote.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// sinLookup (sigProcLib.cpp:177-188) and sinc (:567-571) against the uploaded trig table
// The reference reduces the argument with `while (arg > 1) arg -= 1; while (arg < 0) arg += 1;` -- on a CPU thread a long
// loop, on the GPU a wave that never ends once |arg| >= 2^24 (arg - 1 == arg).  Here the reduction is its CLOSED FORM, no loop:
// below 2^24 every subtraction of the first loop is exact (arg and 1 are multiples of ulp(arg) <= 1 and the difference is
// smaller), so n steps leave arg - n with n = ceil(arg) - 1, one exact subtraction; of the second loop's n = ceil(-arg) steps
// the first n - 1 are exact for the same reason and the last is ONE rounded addition of 1 to a value in [-1, 0) -- and
// fl(arg + n) rounds that same exact sum once.  Same values for every argument the reference's loop ends on; for the others
// (|arg| >= 2^24, infinities, NaN -- the reference never returns) the result is some table entry, the index clamped.
__device__ __forceinline__ float dev_range_reduce(float arg) {
  if (arg > 1.0F) arg = arg - (ceilf(arg) - 1.0F);
  if (arg < 0.0F) arg = arg + ceilf(-arg);
  return arg;
}
__device__ __forceinline__ float dev_sin_lookup(const float *__restrict__ sinT, float x) {
  const float arg = dev_range_reduce(x * (1 / TRX_2PI_F));
  const float argT = arg * (float)TRX_TABLESIZE;
  int argI = (int)argT;
  argI = argI < 0 ? 0 : (argI > TRX_TABLESIZE ? TRX_TABLESIZE : argI);   // (a no-op for every argument the reference's loop ends on)
  const float delta = argT - argI;
  const float iDelta = 1.0F - delta;
  return iDelta * sinT[argI] + delta * sinT[argI + 1];
}
// expjLookup (:192-204) with its subtract-one range reduction bounded: the host entry points refuse phases beyond
// TRX_FSHIFT_MAXPHASE, for which the reference's loop would run for ever (a float that large no longer changes by 1) or for
// a very long time
#define TRX_FSHIFT_MAXPHASE 25000.0f
__device__ __forceinline__ cx dev_expj_lookup(const TrxTables *__restrict__ T, float x) {
  const float arg = dev_range_reduce(x * (1 / TRX_2PI_F));
  const float argT = arg * (float)TRX_TABLESIZE;
  int argI = (int)argT;
  argI = argI < 0 ? 0 : (argI > TRX_TABLESIZE ? TRX_TABLESIZE : argI);   // (only a phase the host refused could get here)
  const float delta = argT - argI;
  const float iDelta = 1.0F - delta;
  return mk(iDelta * T->cosT[argI] + delta * T->cosT[argI + 1], iDelta * T->sinT[argI] + delta * T->sinT[argI + 1]);
}
__device__ __forceinline__ float dev_sinc(const float *__restrict__ sinT, float x) {
  if ((x >= 0.01F) || (x <= -0.01F)) return dev_sin_lookup(sinT, x) / x;
  return 1.0F;
}

}  // namespace
