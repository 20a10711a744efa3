// trxsig_fec.hip -- gfx950 kernel of the GSM L1 FEC soft decode that consumes the burst path's soft
// bits: SoftVector::decode with the rate-1/2, order-4 Viterbi coder
// (CommonLibs/BitVector.cpp:290-524, "bv:"), the Parity/Generator shift registers
// (CommonLibs/BitVector.h:39-112, "bh:"), LSB8MSB + pack (bv:166-195, 541-552) and the XCCH / RACH
// decoder flows (GSM/GSML1FEC.cpp:475-514, 584-653, "fec:"), optionally with the UDP hop's 8-bit
// quantisation in between (Transceiver/Transceiver.cpp:669, TRXManager/TRXManager.cpp:231).
//
// Decomposition: the decoder has 16 survivors, so one block (code word) takes 16 lanes -- lane s IS
// survivor s -- and a wave decodes four blocks.  Measured (MI355X, 16 K XCCH blocks = 64 K bursts):
// 102 us; it is bound by the ~50-instruction dependent chain of a trellis step (4 waves per SIMD), not
// by memory -- the first version, whose whole-block metric table left 2.5 waves per SIMD, took 217 us.  Per trellis step a lane fetches its two predecessors
// (survivors s>>1 and 8 + s>>1: ds_bpermute), adds the branch costs and keeps the cheaper one; the
// first-minimum survivor (bv:382-393) is found with four DPP min exchanges + a ballot and its deferred
// input bit (24 steps back) is the output.  Numerical contract as in trxsig_dev.h: float costs are added
// exactly as the reference adds them (cost + (second-bit cost + first-bit cost), -ffp-contract=off),
// so survivor selection, ties included, is bit-identical.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "trxsig_launch.h"

namespace {

constexpr int kDeferral = 24;                              // 6*mOrder (bh:138)
constexpr int kChunk = 64;                                 // trellis steps per table refill (8 positions per lane)

// coder output for the 5-bit input history idx: generator 0x19 in bit 1, 0x1b in bit 0 (bv:306-330),
// two bits per entry, 32 entries
constexpr unsigned apply_poly(unsigned val, unsigned poly) {
  unsigned prod = val & poly, sum = prod;
  for (unsigned i = 1; i < 5; i++) sum ^= prod >> i;
  return sum & 1u;
}
constexpr unsigned long long gen_table() {
  unsigned long long t = 0;
  for (unsigned idx = 0; idx < 32; idx++)
    t |= (unsigned long long)((apply_poly(idx, 0x19) << 1) | apply_poly(idx, 0x1b)) << (2 * idx);
  return t;
}
constexpr unsigned long long kGen = gen_table();

// The syndrome / parity registers are linear over GF(2) (they start from zero), so the word for a
// bit string is the XOR of the words of its set bits.  XcchSyn::v[i] = syndromeShift response
// (bh:69-74) to a single 1 at position i of a 224-bit code word, generator 0x10004820009 (40 bits);
// RachPar::v[i] = encoderShift response (bh:80-85) to a single 1 at position i of 8 bits, generator 0x6f.
struct XcchSyn {
  unsigned long long v[224];
  constexpr XcchSyn() : v() {
    unsigned long long st = 1;                             // the bit has just been shifted in
    for (int i = 223; i >= 0; i--) {
      v[i] = st & ((1ULL << 40) - 1);
      const unsigned long long fb = (st >> 39) & 1ULL;     // one more zero shifted in behind it
      st <<= 1;
      if (fb) st ^= 0x10004820009ULL;
    }
  }
};
struct RachPar {
  unsigned v[8];
  constexpr RachPar() : v() {
    for (int i = 0; i < 8; i++) {
      unsigned st = 0;
      for (int k = 0; k < 8; k++) {
        const unsigned fb = ((st >> 5) ^ (k == i ? 1u : 0u)) & 1u;
        st <<= 1;
        if (fb) st ^= 0x6fu;
      }
      v[i] = st & 0x3fu;
    }
  }
};
struct TchPar {                                            // encoderShift response, 50 bits, generator 0x0b (3 bits)
  unsigned v[50];
  constexpr TchPar() : v() {
    for (int i = 0; i < 50; i++) {
      unsigned st = 0;
      for (int k = 0; k < 50; k++) {
        const unsigned fb = ((st >> 2) ^ (k == i ? 1u : 0u)) & 1u;
        st <<= 1;
        if (fb) st ^= 0x0bu;
      }
      v[i] = st & 7u;
    }
  }
};
__device__ __constant__ const TchPar kTchPar;
__device__ __constant__ const XcchSyn kXcchSyn;
__device__ __constant__ const RachPar kRachPar;

template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) { return __int_as_float(dpp_i<CTRL>(__float_as_int(v))); }

__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the soft value as the GSM side sees it after the UDP hop
__device__ __forceinline__ float wire_value(float v) {
  const int q = (int)round((double)v * 255.0);             // (char) round(x*255.0), Transceiver.cpp:669
  return (float)(unsigned char)q / 256.0F;                 // TRXManager.cpp:231
}

enum { FEC_GENERIC = 0, FEC_XCCH = 1, FEC_RACH = 2, FEC_TCH = 3 };

// MODE FEC_GENERIC: block b reads soft[b*in_stride + p], p < n, and writes nout bits as bytes to out0 + b*out_stride.
// MODE FEC_XCCH   : block b = bursts 4b..4b+3 of soft[burst*in_stride + 0..147]; c[k] = i[k%4][j(k)] with the
//                   e-bits at 3..59 and 88..144 (fec:607-608, 618-629); out0 = 23 octets per block, out1 = ok.
// MODE FEC_RACH   : block b = burst b, e = burst[49..85) (fec:479); out0 = tail ok, out1 = BSIC, out2 = RA.
// MODE FEC_TCH    : block b = bursts 4b..4b+7 (diagonal deinterleaver, fec:1108-1116), class 1 = c[0..378) decoded,
//                   class 2 = c[378..456) sliced (fec:1133-1163); out0 = d[260] packed MSB first (33 octets),
//                   out1 = good (parity of class 1a and tail), out2 = stolen (Hl of the block's last burst, fec:1077).
// ilv8 (FEC_XCCH only): read c[] through the TCH deinterleaver instead -- the FACCH decode of a stolen block.
template <int MODE>
__global__ __launch_bounds__(64) void k_fec_viterbi(const float *__restrict__ soft, long long in_stride, int n, int nout,
                                                    int nblk, int wire, int ilv8, uint8_t *__restrict__ out0,
                                                    uint8_t *__restrict__ out1, uint8_t *__restrict__ out2,
                                                    long long out_stride) {
  // costs of coder bit 0/1 for both bits of a step, kChunk steps at a time: a small table keeps 8 waves
  // per SIMD resident (the whole 252-step table of an XCCH block would be 4 KB per block: 2.5 waves)
  __shared__ float4 ktab[4][kChunk];
  const int lane = threadIdx.x & 63, row = lane >> 4, s = lane & 15;
  const int blk = blockIdx.x * 4 + row;
  const bool live = blk < nblk;
  const int steps = nout + kDeferral;
  float2 *K2 = reinterpret_cast<float2 *>(ktab[row]);
  const float4 *K = ktab[row];

  float cost = 0.0f;                                       // lane s is survivor s (bv:334-399)
  unsigned ist = 0, outw = 0;
  const int srcA = (lane & 48) + (s >> 1), srcB = srcA + 8;
  for (int c0 = 0; c0 < steps; c0 += kChunk) {
    // ---- metric tables (bv:462-485) for steps c0 .. c0+kChunk-1: positions 2*c0 .. 2*c0 + 2*kChunk - 1,
    //      eight per lane, their loads issued together ----
    float vv[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int p = 2 * c0 + s + 16 * q;
      float v = 0.0f;
      if (p < n && live) {
        if (MODE == FEC_XCCH || MODE == FEC_TCH) {
          const int B = (MODE == FEC_TCH || ilv8) ? (p & 7) : (p & 3);      // burst within the block
          const int j = 2 * ((49 * p) % 57) + ((p % 8) / 4);                // GSM 05.03 4.1.4 / 3.1.3 (fec:622-625, 1111)
          v = soft[(size_t)(4 * blk + B) * in_stride + (j < 57 ? 3 + j : 88 + (j - 57))];
        } else if (MODE == FEC_RACH) {
          v = soft[(size_t)blk * in_stride + 49 + p];
        } else {
          v = soft[(size_t)blk * in_stride + p];
        }
      }
      vv[q] = v;
    }
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int p = 2 * c0 + s + 16 * q;
      float k0 = 0.5F, k1 = 0.5F;                          // past the data: unknowns (bv:481-484)
      if (p < n && live) {
        float v = vv[q];
        if (wire) v = wire_value(v);
        const bool hard = v > 0.5F;                        // sliced() (bv:424-433)
        float pVal = v;
        if (pVal > 0.5F) pVal = 1.0F - pVal;
        float ipVal = 1.0F - pVal;
        if (pVal < 0.01F) pVal = (float)0.01;
        if (ipVal < 0.01F) ipVal = (float)0.01;
        const float match = 0.25F / ipVal, mismatch = 0.25F / pVal;
        k0 = hard ? mismatch : match;                      // coder bit 0 mismatches a received 1
        k1 = hard ? match : mismatch;
      }
      K2[s + 16 * q] = make_float2(k0, k1);
    }
    wave_fence();

    // ---- kChunk trellis steps ----
    const int tend = (c0 + kChunk < steps) ? c0 + kChunk : steps;
    for (int t = c0; t < tend; t++) {
      const float4 k = K[t - c0];                          // {first bit: cost of coder 0, 1; second bit: cost of coder 0, 1}
      const float cA0 = __shfl(cost, srcA, 64), cB0 = __shfl(cost, srcB, 64);
      const unsigned iA = (__shfl(ist, srcA, 64) << 1) | (s & 1), iB = (__shfl(ist, srcB, 64) << 1) | (s & 1);
      const unsigned gA = (unsigned)(kGen >> (2 * (iA & 31))) & 3u, gB = (unsigned)(kGen >> (2 * (iB & 31))) & 3u;
      // cost += cTab[m&1][1] + cTab[(m>>1)&1][0] (bv:365)
      const float cA = cA0 + (((gA & 1u) ? k.w : k.z) + ((gA >> 1) ? k.y : k.x));
      const float cB = cB0 + (((gB & 1u) ? k.w : k.z) + ((gB >> 1) ? k.y : k.x));
      const bool takeA = cA < cB;                          // pruneCandidates (bv:371-379)
      cost = takeA ? cA : cB;
      ist = takeA ? iA : iB;
      if (t >= kDeferral) {
        // minCost (bv:382-393): the FIRST survivor with the minimum cost.  The minimum itself by four DPP
        // exchanges; the lanes that hold it by ballot; the first of them = lowest set bit of the row's mask.
        float mc = cost;
        mc = fminf(mc, dpp_f<0xB1>(mc));                   // quad_perm [1,0,3,2]
        mc = fminf(mc, dpp_f<0x4E>(mc));                   // quad_perm [2,3,0,1]
        mc = fminf(mc, dpp_f<0x141>(mc));                  // row_half_mirror
        mc = fminf(mc, dpp_f<0x140>(mc));                  // row_mirror
        const unsigned long long eq = __builtin_amdgcn_ballot_w64(cost == mc);
        const unsigned long long ob = __builtin_amdgcn_ballot_w64((ist >> kDeferral) & 1u);
        const unsigned rowmask = (unsigned)(eq >> (lane & 48)) & 0xFFFFu;
        const int mi = __builtin_ctz(rowmask | 0x10000u);
        const int op = t - kDeferral;
        const unsigned bit = (unsigned)(ob >> ((lane & 48) + mi)) & 1u;
        if ((op >> 5) == s) outw |= bit << (op & 31);      // lane s collects output bits 32s .. 32s+31
      }
    }
    wave_fence();                                          // table reads done before the next chunk overwrites it
  }
  if (!live) return;

  if (MODE == FEC_GENERIC) {
    for (int k = 0; k < 32; k++) {
      const int i = 32 * s + k;
      if (i < nout) out0[(size_t)blk * out_stride + i] = (uint8_t)((outw >> k) & 1u);
    }
  } else if (MODE == FEC_XCCH) {
    // d[] = u[0..184) with every octet bit-reversed (LSB8MSB, fec:598) and packed MSB first: octet o is
    // u[8o .. 8o+7] with u[8o] as its LSB, i.e. the bytes of the output words as they stand
    for (int q = 0; q < 4; q++)
      if (4 * s + q < 23) out0[(size_t)blk * 23 + 4 * s + q] = (uint8_t)(outw >> (8 * q));
    // syndrome of d[]:~p[] (fec:644-651): XOR of the unit responses of the set bits
    unsigned w = outw;
    if (s == 5) w ^= 0xFF000000u;                          // parity bits 184..223 are inverted
    if (s == 6) w ^= 0xFFFFFFFFu;
    unsigned long long syn = 0;
    if (s < 7) {
      for (int k = 0; k < 32; k++) {
        const unsigned long long r = kXcchSyn.v[32 * s + k];
        if ((w >> k) & 1u) syn ^= r;
      }
    }
    unsigned lo = (unsigned)syn, hi = (unsigned)(syn >> 32);
    lo ^= (unsigned)dpp_i<0xB1>((int)lo); hi ^= (unsigned)dpp_i<0xB1>((int)hi);
    lo ^= (unsigned)dpp_i<0x4E>((int)lo); hi ^= (unsigned)dpp_i<0x4E>((int)hi);
    lo ^= (unsigned)dpp_i<0x141>((int)lo); hi ^= (unsigned)dpp_i<0x141>((int)hi);
    lo ^= (unsigned)dpp_i<0x140>((int)lo); hi ^= (unsigned)dpp_i<0x140>((int)hi);
    if (s == 0) out1[blk] = (lo | hi) == 0;
  } else if (MODE == FEC_TCH) {
    // u[189] -> LDS (word w = bits 32w..32w+31), then d[] (fec:1141-1146) octet by octet
    unsigned *uw = reinterpret_cast<unsigned *>(ktab[row]);
    if (s < 6) uw[s] = outw;
    wave_fence();
    auto ubit = [&](int i) { return (uw[i >> 5] >> (i & 31)) & 1u; };
    auto cbit = [&](int k) {                               // class 2: c[k] sliced, k >= 378
      const int j = 2 * ((49 * k) % 57) + ((k % 8) / 4);
      float v = soft[(size_t)(4 * blk + (k & 7)) * in_stride + (j < 57 ? 3 + j : 88 + (j - 57))];
      if (wire) v = wire_value(v);
      return v > 0.5F ? 1u : 0u;
    };
    auto dbit = [&](int q) -> unsigned {
      if (q >= 260) return 0u;
      if (q >= 182) return cbit(378 + q - 182);
      const int k = q >> 1;
      return (q & 1) ? ubit(184 - k) : ubit(k);
    };
    for (int o = s; o < 33; o += 16) {
      unsigned byte = 0;
      for (int q = 0; q < 8; q++) byte = (byte << 1) | dbit(8 * o + q);
      out0[(size_t)blk * 33 + o] = (uint8_t)byte;
    }
    if (s == 0) {
      unsigned calc = 0;
      for (int i = 0; i < 50; i++) if (dbit(i)) calc ^= kTchPar.v[i];
      const unsigned sent = (~((ubit(91) << 2) | (ubit(92) << 1) | ubit(93))) & 7u;      // peekField(91,3)
      const unsigned tail = ubit(185) | ubit(186) | ubit(187) | ubit(188);
      out1[blk] = (sent == calc) && (tail == 0);
      float hl = soft[(size_t)(4 * blk + 7) * in_stride + 60];
      if (wire) hl = wire_value(hl);
      out2[blk] = hl > 0.5F;
    }
  } else {
    if (s == 0) {
      const unsigned u = outw;                             // bit k = u[k]
      const bool tail_ok = ((u >> 14) & 0xFu) == 0;        // fec:485
      unsigned sent = 0, chk = 0;
      for (int k = 0; k < 6; k++) sent |= ((u >> (8 + k)) & 1u) << (5 - k);     // peekField(8,6), MSB first
      for (int i = 0; i < 8; i++) if ((u >> i) & 1u) chk ^= kRachPar.v[i];
      out0[blk] = tail_ok;
      out1[blk] = (uint8_t)((~sent ^ chk) & 0x3fu);        // fec:490-493
      out2[blk] = (uint8_t)(u & 0xFFu);                    // RA = d[] after LSB8MSB, MSB first (fec:506-507)
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_fec_xcch_encode: XCCHL1Encoder::sendFrame / encode / interleave / transmit (fec:772-845), one wave per L2
// frame: d[] = the 23 octets LSB first (LSB8MSB, fec:789), 40 parity bits = ~(Fire-code remainder of d[])
// (writeParityWord, bv:409-416), four zero tail bits, rate-1/2 coder (BitVector::encode, bv:217-239),
// GSM 05.03 4.1.4 interleaver into the e-bits of four bursts, plus what the encoder's constructor puts in
// every burst: zero tails, both stealing flags set (fec:713-717) and the training sequence at 61..86.
// All integer work; the parity word uses the register's GF(2) linearity (XOR of unit responses).
// ---------------------------------------------------------------------------------------------
struct XcchPar {                                           // encoderShift response (bh:80-85) to a 1 at position i of 184
  unsigned long long v[184];
  constexpr XcchPar() : v() {
    unsigned long long st = 0x10004820009ULL & ((1ULL << 40) - 1);   // the bit just went in: fb = 1, state = coeff
    for (int i = 183; i >= 0; i--) {
      v[i] = st & ((1ULL << 40) - 1);
      const unsigned long long fb = (st >> 39) & 1ULL;     // one more zero behind it
      st <<= 1;
      if (fb) st ^= 0x10004820009ULL;
    }
  }
};
__device__ __constant__ const XcchPar kXcchPar;

__global__ __launch_bounds__(64) void k_fec_xcch_encode(const uint8_t *__restrict__ frames, int nblk,
                                                        const uint8_t *__restrict__ tsc_bits /* 26 */, uint8_t *__restrict__ bits) {
  __shared__ unsigned uw[8];                                // u[228], bit k of word w = u[32w + k]
  const int lane = threadIdx.x, blk = blockIdx.x;
  if (blk >= nblk) return;
  // d[]: octet o contributes u[8o + m] = bit m of the octet (LSB first)
  unsigned w = 0;
  if (lane < 6)
    for (int q = 0; q < 4; q++)
      if (4 * lane + q < 23) w |= (unsigned)frames[(size_t)blk * 23 + 4 * lane + q] << (8 * q);
  // parity word of d[0..184): every lane XORs the unit responses of its word's set bits
  unsigned long long par = 0;
  if (lane < 6)
    for (int k = 0; k < 32; k++)
      if (32 * lane + k < 184 && ((w >> k) & 1u)) par ^= kXcchPar.v[32 * lane + k];
  unsigned lo = (unsigned)par, hi = (unsigned)(par >> 32);
  for (int m = 1; m < 8; m <<= 1) { lo ^= __shfl_xor(lo, m, 64); hi ^= __shfl_xor(hi, m, 64); }
  const unsigned long long pw = ~(((unsigned long long)hi << 32) | lo) & ((1ULL << 40) - 1);   // inverted (bv:413)
  // u[184 + k] = bit (39 - k) of the word (fillField, MSB first); u[224..227] = 0
  if (lane == 5) for (int k = 0; k < 8; k++) w |= (unsigned)((pw >> (39 - k)) & 1ULL) << (24 + k);
  if (lane == 6) { w = 0; for (int k = 0; k < 32; k++) w |= (unsigned)((pw >> (39 - 8 - k)) & 1ULL) << k; }
  if (lane == 7) w = 0;
  if (lane < 8) uw[lane] = w;
  wave_fence();
  auto ubit = [&](int i) { return i < 0 ? 0u : ((uw[i >> 5] >> (i & 31)) & 1u); };
  uint8_t *out = bits + (size_t)blk * 4 * 148;
  // burst skeleton: zeros, stealing flags, training sequence
  for (int i = lane; i < 4 * 148; i += 64) {
    const int p = i % 148;
    uint8_t v = 0;
    if (p == 60 || p == 87) v = 1;
    else if (p >= 61 && p < 87) v = tsc_bits[p - 61] & 1u;
    if (p < 3 || p >= 145 || (p >= 60 && p < 88)) out[i] = v;
  }
  // c[2k], c[2k+1] from the 5-bit history ending at u[k]; interleave and place (fec:812-817, 833-836)
  for (int k = lane; k < 228; k += 64) {
    const unsigned idx = ubit(k) | (ubit(k - 1) << 1) | (ubit(k - 2) << 2) | (ubit(k - 3) << 3) | (ubit(k - 4) << 4);
    const unsigned g = (unsigned)(kGen >> (2 * idx)) & 3u;
    for (int h = 0; h < 2; h++) {
      const int c = 2 * k + h;
      const int B = c & 3, j = 2 * ((49 * c) % 57) + ((c % 8) / 4);
      out[B * 148 + (j < 57 ? 3 + j : 88 + (j - 57))] = (uint8_t)(h == 0 ? (g >> 1) : (g & 1u));
    }
  }
}

}  // namespace

hipError_t trx_launch_fec_xcch_encode(hipStream_t st, const uint8_t *frames, int nblk, const uint8_t *tsc_bits, uint8_t *bits,
                                      TrxProfiler *prof) {
  if (nblk <= 0) return hipSuccess;
  if (prof) prof->begin(TRXSIG_K_FEC, st);
  k_fec_xcch_encode<<<dim3(nblk), dim3(64), 0, st>>>(frames, nblk, tsc_bits, bits);
  if (prof) prof->end(TRXSIG_K_FEC, st);
  return hipGetLastError();
}

hipError_t trx_launch_fec(hipStream_t st, int mode, const float *soft, long long in_stride, int n, int nout, int nblk,
                          int wire, uint8_t *out0, uint8_t *out1, uint8_t *out2, long long out_stride, TrxProfiler *prof,
                          int ilv8) {
  if (nblk <= 0) return hipSuccess;
  if (nout <= 0 || nout > 512 || n < 0 || n > 2 * nout) return hipErrorInvalidValue;
  const dim3 grid((nblk + 3) / 4), block(64);
  if (prof) prof->begin(TRXSIG_K_FEC, st);
  switch (mode) {
    case FEC_GENERIC: k_fec_viterbi<FEC_GENERIC><<<grid, block, 0, st>>>(soft, in_stride, n, nout, nblk, wire, ilv8, out0, out1, out2, out_stride); break;
    case FEC_XCCH: k_fec_viterbi<FEC_XCCH><<<grid, block, 0, st>>>(soft, in_stride, n, nout, nblk, wire, ilv8, out0, out1, out2, out_stride); break;
    case FEC_RACH: k_fec_viterbi<FEC_RACH><<<grid, block, 0, st>>>(soft, in_stride, n, nout, nblk, wire, ilv8, out0, out1, out2, out_stride); break;
    case FEC_TCH: k_fec_viterbi<FEC_TCH><<<grid, block, 0, st>>>(soft, in_stride, n, nout, nblk, wire, ilv8, out0, out1, out2, out_stride); break;
    default: return hipErrorInvalidValue;
  }
  if (prof) prof->end(TRXSIG_K_FEC, st);
  return hipGetLastError();
}
