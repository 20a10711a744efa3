// trxsig_corr.h -- the midamble correlator shared by k_tsc_corr (trxsig_normal.hip) and k_normal_quad
// (trxsig_fused.hip): geometry, and the load / LDS-stage / energy / correlate / argmax round for the four
// bursts of a wave.
// Numerical contract: see trxsig_dev.h / DESIGN.md (every float32 operation is the reference's, in the
// reference's order; built with -ffp-contract=off).
#pragma once
#include "trxsig_dev.h"

namespace {

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
  // the detect -> peak record in memory (round 5): slot PAIRS -- float4 number j * Bpad + b holds slots 2j and 2j + 1 of burst
  // b (slot NS = {M, energy}) -- so that k_tsc_peak2 fetches its 23-lag window and a side of the valley in 12 + 8 sixteen-byte
  // loads instead of 38 eight-byte ones (a wave's load costs the vector memory pipe the same sixteen steps either way)
  static constexpr int NPAIR = (NS + 2) / 2;
};
// slot s of burst b in the paired record (an eight-byte access: the peak kernels of the tuning build, the meta slot)
template <typename T>
__device__ __forceinline__ T &rec_slot(T *rec, int Bpad, int s, int b) { return rec[2 * ((size_t)(s >> 1) * Bpad + b) + (s & 1)]; }

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
  typedef CorrGeom<SPS> G;
  static constexpr int NW = (CorrGeom<SPS>::NL + 15) / 16;     // window samples per lane
  cx w[NW];
  cx e[CorrGeom<SPS>::NEQ];
  int b;
  bool live, good;
  // lane r holds window samples r, r + 16, .. and energy-window samples likewise: into the row / the norms' place
  __device__ __forceinline__ void stage_window(cx *W, int r) const {
#pragma unroll
    for (int i = 0; i < NW; i++) {
      const int q = r + 16 * i;
      if (q < G::NL) W[G::FRONT + q] = w[i];
    }
  }
  __device__ __forceinline__ void stage_norms(float *ef, int r) const {
#pragma unroll
    for (int q = 0; q < G::NEQ; q++) {
      const int i = r + 16 * q;
      if (i < G::NE) ef[i] = norm2(e[q]);
    }
  }
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
template <int SPS, bool REC, bool EFIRST = false, unsigned TAPCLS = TRX_TAPS_GENERIC, typename IN = CorrIn<SPS>>
__device__ __forceinline__ void corr_round(const IN &in, cx *W, float4 *E, int lane, int r, const cx (&tap)[16],
                                           cx *__restrict__ rec, int Bpad, int &M_out, float &energy_out) {
  typedef CorrGeom<SPS> G;
  auto stage_norms = [&] { in.stage_norms(reinterpret_cast<float *>(E), r); };
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
  in.stage_window(W, r);
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
  // ---- record: corr[M-H .. M+H] (zeros outside [0,NL)), then {M, energy}; a lane writes slot pairs (one 16-byte store each) ----
  if (REC && in.live) {
    const int M = bestT;
    auto slot = [&](int s) {
      cx v = mk(0, 0);
      if (s < G::NS) {
        const int lag = M - G::H + s;
        if (lag >= 0 && lag < G::NL) v = W[lag];
      } else if (s == G::NS) {
        v = mk(__int_as_float(in.good ? M : -2), energy);  // M = -2 marks an invalid burst
      }
      return v;
    };
    for (int j = r; j < G::NPAIR; j += 16) {
      const cx a = slot(2 * j), c = slot(2 * j + 1);
      reinterpret_cast<float4 *>(rec)[(size_t)j * Bpad + in.b] = make_float4(a.r, a.i, c.r, c.i);
    }
  }
  wave_lds_fence();                                        // record reads done before the row is reused
  (void)lane;
}


// class of every tap (see TapPattern): 1 = real part exactly +-1, 2 = imaginary part exactly +-1, 0 = neither
inline unsigned tap_classes(const TrxTables *hT, int tsc) {
  unsigned m = 0;
  for (int k = 0; k < 16; k++) {
    const trx_c32 a = hT->mid_ctap[tsc][k];
    const unsigned c = (a.r == 1.0f || a.r == -1.0f) ? 1u : ((a.i == 1.0f || a.i == -1.0f) ? 2u : 0u);
    m |= c << (2 * k);
  }
  return m;
}

}  // namespace
