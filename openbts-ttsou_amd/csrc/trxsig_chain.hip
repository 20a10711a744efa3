// trxsig_chain.hip -- the normal-burst leg as ONE launch whose workgroups hand over inside the launch.
//
//   k_normal_chain: every workgroup (256 threads) has one of two roles, fixed by its index:
//     Q  "detect"      16 bursts, 16 lanes (one DPP row) per burst: energyDetect, the midamble correlation with
//                      the 16 non-zero taps, argmax, peakDetect's bisection speculated three levels at a time,
//                      analyzeTrafficBurst's tail (trxsig_corr.h / trxsig_bisect.h: k_tsc_corr's and the
//                      speculative peak code).  Writes flags / amp / TOA / avgPwr and publishes a 16-byte
//                      granule {amp.re, amp.im, toa, tag | flags} per burst (one write-through store).
//     D  "demodulate"  4 bursts, a wave per burst: requests the burst's samples, then reads the granule of its
//                      burst until the tag is there, clears the tag, and runs demodulateBurst (fused_demod,
//                      k_demod's arithmetic).
//   Why: as three launches the leg serialises a VALU-bound phase (correlation), a latency-bound phase (bisection)
//   and an HBM-bound phase (demodulation), moves 1.5x the algorithmic bytes (window read twice from HBM, a 352-byte
//   record per burst written and read back) and pays ramp-up and drain three times.  In one launch the detect
//   workgroups of later bursts compute while the demodulate workgroups of earlier bursts wait for memory; the
//   window the demodulator re-reads and the granule are a few microseconds old (L2 / memory-side cache), and the
//   record is gone.
//
//   Order.  Workgroup index i belongs to stream x = i % 8 (round-robin dispatch puts a stream on one XCD -- that is
//   for L2 locality only) at position k = i / 8; stream x owns the 16-burst tiles t = 8*lt + x.  Within a stream
//   the positions are: Q(0) .. Q(Lg-1), then groups {Q(Lg+g), D(g,0..3)}, then the D of the last Lg tiles.  So the
//   detect workgroup of a tile always sits Lg tiles (5*Lg positions) ahead of the tile's demodulators, and a
//   demodulator only ever waits for a workgroup with a smaller index.  Workgroups are started in index order, so
//   that workgroup is running or done: no wait can be circular.  HIP does not promise that order; therefore every
//   wait is bounded -- a demodulator that gives up raises *status (host-visible), the library reports the call as
//   failed at its next entry and goes back to the three-launch path for good (trxsig_api.cpp).
//
//   Hand-over (MI355X: the L2s of the 8 XCDs are not coherent, a CU's L1 is never refreshed): the granule is the
//   flag -- one aligned 16-byte `global_store_dwordx4 sc1` (write-through) by one lane, read by `global_load_dwordx4
//   sc1` (bypasses L1) until bit 31 of its last word is set.  Nothing else is handed over inside the launch.  The
//   demodulator clears the tag word (agent-scope store) after reading it, so every launch starts with all tags
//   clear and no per-launch epoch or memset is needed (a hipGraph replay of the launch works the same).
//
// Numerical contract: see trxsig_dev.h / DESIGN.md (every float32 operation is the reference's, in the
// reference's order; built with -ffp-contract=off).  Results are bit-identical to the three-launch path.
#include "trxsig_bisect.h"
#include "trxsig_corr.h"
#include "trxsig_demod.h"

#ifndef TRX_CHAIN_DTILE
#define TRX_CHAIN_DTILE 16                                 // bursts per demodulate workgroup: 4 (a burst per wave), 8 or 16
#endif
#define TRX_CHAIN_NDW (16 / TRX_CHAIN_DTILE)               // demodulate workgroups per 16-burst tile
#ifndef TRX_CHAIN_WPS
#define TRX_CHAIN_WPS 6                                    // waves per SIMD the register allocation must allow
#endif

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define TRX_CHAIN_TAG 0x80000000u

__device__ __forceinline__ u32x4 granule_load(const u32x4 *p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void granule_store(u32x4 *p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}

template <int SPS>
struct ChainGeom {
  typedef CorrGeom<SPS> G;
  typedef FusedGeom<SPS, 64> F;
  typedef typename F::D D;
  // a detect row: [0, NL) correlation, then loc[26], then the valley scratch; what is left of the row past that
  // (SLACK ..) holds two rows of the coarse sinc table when they fit (sps 4), so that the table costs no LDS
  static constexpr int SLACK = (G::NL + 26 + F::NV / 2 + 1 + 1) & ~1;
  static constexpr bool STAB_IN_ROWS = (G::WPAD - SLACK) >= 24;
  static constexpr int ROWS_B = 16 * G::WPAD * 8;
  static constexpr int Q_B = ROWS_B + (STAB_IN_ROWS ? 0 : 32 * 24 * 4);
  static constexpr int D_B = 4 * D::U * 8;
  static constexpr int LDS_B = ((Q_B > D_B ? Q_B : D_B) + 15) & ~15;
};

template <int SPS, unsigned TAPCLS, bool TOL = false>
__global__ __launch_bounds__(256, TRX_CHAIN_WPS) void k_normal_chain(
    const TrxTables *__restrict__ T, const cx *__restrict__ samples, const int32_t *__restrict__ offset,
    const int32_t *__restrict__ length, int B, TapArg taps, cx gain_inv, float mid_toa, float detect_thresh,
    float energy_thresh, uint8_t *__restrict__ flags, cx *__restrict__ amp_out, float *__restrict__ toa_out,
    float *__restrict__ avgpwr_out, float *__restrict__ soft, uint8_t *__restrict__ hard, int nsoft, int stride,
    u32x4 *det, unsigned *status, int LT, int Lg, unsigned spin_limit, int dbg) {
  typedef CorrGeom<SPS> G;
  typedef ChainGeom<SPS> CG;
  typedef FusedGeom<SPS, 64> F;
  typedef typename F::D D;
  static_assert(G::WPAD - G::NL >= 26 + F::NV / 2 + 1, "row has no room for the bisection scratch");
  static_assert(8 * G::WPAD >= 4 * G::NE, "the energy norms are staged in the row itself");
  __shared__ __attribute__((aligned(16))) char lds[CG::LDS_B];

  // ---- role and tile of this workgroup (uniform) ----
  const int x = blockIdx.x & 7, k = blockIdx.x >> 3;
  constexpr int NDW = TRX_CHAIN_NDW, GRP = NDW + 1;
  const int mid = GRP * (LT - Lg);
  bool isq;
  int lt, dsub = 0;
  if (k < Lg) { isq = true; lt = k; }
  else if (k < Lg + mid) {
    const int g = (k - Lg) / GRP, j = (k - Lg) - GRP * g;
    if (j == 0) { isq = true; lt = Lg + g; } else { isq = false; lt = g; dsub = j - 1; }
  } else {
    const int m = k - Lg - mid;
    isq = false; lt = LT - Lg + m / NDW; dsub = m % NDW;
  }
  const int b0 = (lt * 8 + x) * 16;
  if (b0 >= B) return;
  if (dbg && (isq ? (dbg & 1) : (dbg & 2))) return;      // timing experiments: one role only
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

  if (isq) {
    // =========================== Q: detect 16 bursts ===========================
    cx(*rows)[G::WPAD] = reinterpret_cast<cx(*)[G::WPAD]>(lds);
    const int row = lane >> 4, r = lane & 15;
    const int slot = wave * 4 + row;
    cx *W = rows[slot];
    // coarse sinc rows f = 0, 16, .., 496 (all that the first two super-steps of the bisection can ask for): this
    // wave fetches the 8 rows that live with its 4 bursts; stored after the correlation (the slack is window padding
    // until then)
    float tv[3];
#pragma unroll
    for (int q = 0; q < 3; q++) { const int ix = lane * 3 + q; tv[q] = T->sinc_grid[16 * (wave * 8 + ix / 24)][ix % 24]; }

    int M;
    float energy;
    CorrIn<SPS> in;
    {
      cx tap[16];
#pragma unroll
      for (int q = 0; q < 16; q++) tap[q] = mk(taps.v[2 * q], taps.v[2 * q + 1]);
      corr_issue<SPS>(in, b0 + slot, B, r, samples, offset, length);
      corr_round<SPS, false, true, TAPCLS>(in, W, reinterpret_cast<float4 *>(W), lane, r, tap, nullptr, 0, M, energy);
    }
    const bool live = in.live, good = in.good;
    const int b = in.b;
    float *stab_sep = reinterpret_cast<float *>(lds + CG::ROWS_B);            // (only when !STAB_IN_ROWS)
    auto stab_row = [&](int s) -> const float4 * {
      if (CG::STAB_IN_ROWS) return reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(rows[s >> 1] + CG::SLACK) + 24 * (s & 1));
      return reinterpret_cast<const float4 *>(stab_sep + 24 * s);
    };
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const int ix = lane * 3 + q, s = wave * 8 + ix / 24;
      const_cast<float *>(reinterpret_cast<const float *>(stab_row(s)))[ix % 24] = tv[q];
    }
    __syncthreads();                                       // the sinc rows of all four waves are in place (the only barrier)

    cx *loc = W + G::NL;                                   // lags M-12 .. M+11 as interpolatePoint sees them (:646)
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
    int e = 0;                                             // early = M-1 + e/512
    asm volatile("" : "+v"(e));
    bool active = true;
    cx peak = mk(0, 0);
    {
      const int rel = kFusedRel3.v[r];
#pragma unroll
      for (int st = 0; st < 3; st++) {                     // increments 256,128,64 | 32,16,8 | 4,2,1
        const int inc_last = 64 >> (3 * st);
        const int el = e + (rel >> 2) * inc_last;
        float srow[24];
        if (st < 2) {                                      // nodes on multiples of 16/512: the LDS copy
          const float4 *rw = stab_row((el & 511) >> 4);
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
      float srow[24];                                      // interpolatePoint(early + 1) where the loop stopped (:699-700)
      fused_row(T, e, srow);
      peak = fused_point(loc, e, 1, srow);
      asm volatile("" : "+v"(peak.r), "+v"(peak.i));
    }
    cx amp;
    float toa;
    bool detected, energy_ok;
    fused_tail<SPS, 16>([&](int lag) { return (lag < 0 || lag >= G::NL) ? 0.0f : norm2(W[lag]); }, V, r, M, e, peak, good,
                        energy, gain_inv, mid_toa, detect_thresh, energy_thresh, amp, toa, detected, energy_ok);
    if (live && r == 0) {
      unsigned fl = 0;
      if (!good) fl = TRXSIG_F_BADLEN;
      else fl = (energy_ok ? TRXSIG_F_ENERGY : 0) | (detected ? TRXSIG_F_DETECT : 0);
      u32x4 gr;
      gr.x = __float_as_uint(amp.r); gr.y = __float_as_uint(amp.i); gr.z = __float_as_uint(toa); gr.w = TRX_CHAIN_TAG | fl;
      granule_store(det + b, gr);                          // the hand-over: first, so that the demodulators see it soonest
      flags[b] = (uint8_t)fl;
      amp_out[b] = amp;
      toa_out[b] = toa;
      if (avgpwr_out) avgpwr_out[b] = good ? energy / (float)(unsigned)G::NE : 0.0f;
    }
    return;
  }

  // =========================== D: demodulate 16 bursts, four per wave in turn ===========================
  // A wave always has its NEXT burst's samples in flight while it works on the current one: the bytes a wave keeps
  // outstanding, not the number of waves, is what lets this role share the CU with the detect role.
  cx *P = reinterpret_cast<cx *>(lds) + wave * D::U;
  constexpr int NLD = (157 * SPS / 2 + 63) / 64;
  constexpr int BPW = TRX_CHAIN_DTILE / 4;                 // bursts per wave
  const int bw0 = b0 + dsub * TRX_CHAIN_DTILE + wave * BPW;
  auto fetch = [&](int bb, int ln, float4 (&v)[NLD], int &off, int &N, bool &geom) {
    off = 0; N = 0; geom = false;
    if (bb < B) { off = offset[bb]; N = length[bb]; }
    geom = (bb < B) && (off >= 0) && (N >= 92 * SPS) && (N <= 157 * SPS) && (N % SPS == 0);
    const float4 *xv = reinterpret_cast<const float4 *>(samples + (geom ? off : 0));
    const bool wide = (off & 1) == 0;
#pragma unroll
    for (int i = 0; i < NLD; i++) {
      const int q = ln + 64 * i;
      v[i] = (geom && wide && q < N / 2) ? xv[q] : make_float4(0, 0, 0, 0);
    }
  };
  float4 v[NLD];
  int off, N;
  bool geom;
  fetch(bw0, lane, v, off, N, geom);
#pragma unroll 1
  for (int rr = 0; rr < BPW; rr++) {
    int ln = lane;                                         // (opaque copy: keeps per-lane addresses from being hoisted out of the loop)
    asm volatile("" : "+v"(ln));
    const int b = bw0 + rr;                                // wave-uniform
    if (b >= B) break;
    float4 vn[NLD];
    int offn = 0, Nn = 0;
    bool geomn = false;
    if (rr + 1 < BPW) fetch(b + 1, ln, vn, offn, Nn, geomn);
    float *sb = soft + (size_t)b * stride;
    uint8_t *hb = hard ? hard + (size_t)b * stride : nullptr;
    unsigned gw;
    cx amp;
    float toa;
    {
      const u32x4 *dp = det + b;
      unsigned spins = 0;
      for (;;) {
        const u32x4 g = granule_load(dp);
        gw = __builtin_amdgcn_readfirstlane(g.w);
        if (gw & TRX_CHAIN_TAG) {
          amp = mk(__uint_as_float(__builtin_amdgcn_readfirstlane(g.x)), __uint_as_float(__builtin_amdgcn_readfirstlane(g.y)));
          toa = __uint_as_float(__builtin_amdgcn_readfirstlane(g.z));
          break;
        }
        if (++spins > spin_limit) {                        // never seen; see the header comment
          if (ln == 0) { __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
          return;
        }
        __builtin_amdgcn_s_sleep(8);
      }
      if (ln == 0 && !dbg)                                 // tag consumed: the next launch starts from a clear word
        __hip_atomic_store(reinterpret_cast<unsigned *>(det + b) + 3, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const bool enabled = geom && (gw & TRXSIG_F_DETECT) && (fabsf(toa) <= 4096.0f);   // k_demod's gate
    if (!enabled) {
      for (int m = ln; m < nsoft; m += 64) { sb[m] = 0.0f; if (hb) hb[m] = 0; }
    } else {
      const bool wide = (off & 1) == 0;
      if (wide && (N & 1) == 0) {
        if (!(TOL && fused_demod_tol<SPS>(T, P, v, N, amp, toa, ln, sb, hb, nsoft)))   // (TOL: TRXSIG_SOFT_TOLERANCE)
          fused_demod<SPS, 64>(T, P, v, N, amp, toa, ln, sb, hb, nsoft, [] {}, nullptr, nullptr);
      } else demod_core<SPS, false, 148>(T, P, samples, off, N, wide, v, amp, toa, ln, sb, hb, nullptr, nsoft);
      wave_lds_fence();                                    // staging reads done before the next burst overwrites it
    }
#pragma unroll
    for (int i = 0; i < NLD; i++) v[i] = vn[i];
    off = offn; N = Nn; geom = geomn;
  }
}

}  // namespace

size_t trx_chain_ws_bytes(int bursts) { return 16 * (size_t)bursts; }

template <int S>
static void launch_chain(hipStream_t st, const TrxTables *dT, const TrxTables *hT, const trx_c32 *samples, const int32_t *off,
                         const int32_t *len, int B, int tsc, float detect_thresh, float energy_thresh, uint8_t *flags,
                         trx_c32 *amp, float *toa, float *avgpwr, float *soft, uint8_t *hard, int nsoft, int stride,
                         void *det, unsigned *status, int lag, unsigned spin_limit, int generic_taps, int dbg, int tol) {
  TapArg ta;
  for (int k = 0; k < 16; k++) { ta.v[2 * k] = hT->mid_ctap[tsc][k].r; ta.v[2 * k + 1] = hT->mid_ctap[tsc][k].i; }
  // gain.inv() (Complex.h:154-160) in the reference's float arithmetic; this file is built with -ffp-contract=off
  const trx_c32 g = hT->mid_gain[tsc];
  const float n = g.i * g.i + g.r * g.r;
  trx_c32 ginv; ginv.r = g.r / n; ginv.i = -g.i / n;
  const int NT = (B + 15) / 16, LT = (NT + 7) / 8;
  const int Lg = lag < 1 ? 1 : (lag > LT ? LT : lag);
  const dim3 grid(8 * (TRX_CHAIN_NDW + 1) * LT), block(256);
#define TRX_CHAIN_ARGS dT, samples, off, len, B, ta, ginv, hT->mid_toa[tsc], detect_thresh, energy_thresh, flags, amp, toa, \
                       avgpwr, soft, hard, nsoft, stride, (u32x4 *)det, status, LT, Lg, spin_limit, dbg
  if (tol && !generic_taps && tap_classes(hT, tsc) == TapPattern<S>::value)
    k_normal_chain<S, TapPattern<S>::value, true><<<grid, block, 0, st>>>(TRX_CHAIN_ARGS);
  else if (tol)
    k_normal_chain<S, TRX_TAPS_GENERIC, true><<<grid, block, 0, st>>>(TRX_CHAIN_ARGS);
  else if (!generic_taps && tap_classes(hT, tsc) == TapPattern<S>::value)
    k_normal_chain<S, TapPattern<S>::value><<<grid, block, 0, st>>>(TRX_CHAIN_ARGS);
  else
    k_normal_chain<S, TRX_TAPS_GENERIC><<<grid, block, 0, st>>>(TRX_CHAIN_ARGS);
#undef TRX_CHAIN_ARGS
}

hipError_t trx_launch_normal_chain(hipStream_t st, int sps, const TrxTables *dT, const TrxTables *hT, const trx_c32 *samples,
                                   const int32_t *off, const int32_t *len, int B, int tsc, float detect_thresh,
                                   float energy_thresh, uint8_t *flags, trx_c32 *amp, float *toa, float *avgpwr, float *soft,
                                   uint8_t *hard, int nsoft, int stride, void *det, unsigned *status, int lag,
                                   unsigned spin_limit, int generic_taps, TrxProfiler *prof, int dbg, int soft_tolerance) {
  if (B <= 0) return hipSuccess;
  if (nsoft <= 0 || nsoft > 148 || !det || !status) return hipErrorInvalidValue;
  if (prof) prof->begin(TRXSIG_K_NORMAL_CHAIN, st);
  switch (sps) {
    case 1: launch_chain<1>(st, dT, hT, samples, off, len, B, tsc, detect_thresh, energy_thresh, flags, amp, toa, avgpwr, soft, hard, nsoft, stride, det, status, lag, spin_limit, generic_taps, dbg, soft_tolerance); break;
    case 2: launch_chain<2>(st, dT, hT, samples, off, len, B, tsc, detect_thresh, energy_thresh, flags, amp, toa, avgpwr, soft, hard, nsoft, stride, det, status, lag, spin_limit, generic_taps, dbg, soft_tolerance); break;
    case 4: launch_chain<4>(st, dT, hT, samples, off, len, B, tsc, detect_thresh, energy_thresh, flags, amp, toa, avgpwr, soft, hard, nsoft, stride, det, status, lag, spin_limit, generic_taps, dbg, soft_tolerance); break;
    default: return hipErrorInvalidValue;
  }
  if (prof) prof->end(TRXSIG_K_NORMAL_CHAIN, st);
  return hipGetLastError();
}
