/*
 * oracle/fec_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * CPU restatement (plain C99) of the GSM L1 FEC soft decode that consumes the burst path's soft bits
 * (SURVEY 8f rank 1): the rate-1/2, order-4 Viterbi decoder with deferral 24
 * (CommonLibs/BitVector.cpp:290-524, "bv:<line>"), the Generator/Parity shift registers
 * (CommonLibs/BitVector.h:39-112, "bh:<line>"), LSB8MSB (bv:166-195) and the XCCH / RACH decoder flows
 * (GSM/GSML1FEC.cpp:475-514, 584-653, "fec:<line>"), plus the UDP wire quantisation that sits between
 * the transceiver and the decoder (Transceiver/Transceiver.cpp:669, TRXManager/TRXManager.cpp:231).
 *
 * Parity status: PINNED -- checked bit for bit against the real reference compiled in place
 * (oracle/_ref/libref_fec.so, tests/test_fec_oracle.py, build container only), against the
 * known-answer input of CommonLibs/BitVectorTest.cpp:72 and the golden vectors captured from that
 * reference (tests/golden/fec.npz, everywhere).  Float costs are accumulated exactly as the
 * reference does (one float add per candidate per step, -ffp-contract=off).
 */
#include "fec_oracle.h"

#include <math.h>
#include <omp.h>
#include <string.h>

#define FO_DEFERRAL 24      /* 6*mOrder, bh:138 */

/* applyPoly (bv:41-47) over order+1 = 5 taps; generator table (bv:306-330): coder output for the 5-bit
   input history `index`, generator 0x19 in bit 1 and 0x1b in bit 0 */
static unsigned apply_poly(unsigned val, unsigned poly, unsigned order) {
  const unsigned prod = val & poly;
  unsigned sum = prod;
  for (unsigned i = 1; i < order; i++) sum ^= prod >> i;
  return sum & 1u;
}
static void gen_table(unsigned g[32]) {
  for (unsigned index = 0; index < 32; index++)
    g[index] = (apply_poly(index, 0x19, 5) << 1) | apply_poly(index, 0x1b, 5);
}

/* BitVector::encode (bv:217-239) */
void fo_encode(const uint8_t *bits, int n, uint8_t *out) {
  unsigned accum = 0;
  for (int i = 0; i < n; i++) {
    accum = (accum << 1) | (bits[i] & 1u);
    const unsigned index = accum & 0x1f;
    out[2 * i] = (uint8_t)apply_poly(index, 0x19, 5);
    out[2 * i + 1] = (uint8_t)apply_poly(index, 0x1b, 5);
  }
}

/* SoftVector::decode (bv:438-524) with ViterbiR2O4::step (bv:334-399) */
void fo_viterbi_decode(const float *soft, int n, uint8_t *out, int nout) {
  enum { MAXN = 1024 };
  unsigned g[32];
  gen_table(g);
  const int ctsz = n + 2 * FO_DEFERRAL;
  uint32_t history[MAXN + 2 * FO_DEFERRAL];
  float match[MAXN + 2 * FO_DEFERRAL], mismatch[MAXN + 2 * FO_DEFERRAL];
  if (n > MAXN) n = MAXN;
  {
    uint32_t accum = 0;
    for (int i = 0; i < n; i++) {                           /* sliced(): > 0.5F (bv:424-433) */
      accum = (accum << 1) | (soft[i] > 0.5F ? 1u : 0u);
      history[i] = accum;
    }
    for (int i = n; i < ctsz; i++) {                        /* repeat the last bit (bv:457-460) */
      accum = (accum << 1) | (accum & 1u);
      history[i] = accum;
    }
  }
  for (int i = 0; i < n; i++) {                             /* bv:467-478 */
    float pVal = soft[i];
    if (pVal > 0.5F) pVal = 1.0F - pVal;
    float ipVal = 1.0F - pVal;
    if (pVal < 0.01F) pVal = (float)0.01;
    if (ipVal < 0.01F) ipVal = (float)0.01;
    match[i] = 0.25F / ipVal;
    mismatch[i] = 0.25F / pVal;
  }
  for (int i = n; i < ctsz; i++) { match[i] = 0.5F; mismatch[i] = 0.5F; }

  struct { uint32_t iState, oState; float cost; } surv[16], cand[32];
  memset(surv, 0, sizeof surv);
  memset(cand, 0, sizeof cand);
  int ip = 1, oc = 0, op = 0;                               /* ip = history + step - 1 (bv:494) */
  const float *mt = match, *mm = mismatch;
  while (op < nout) {
    /* branchCandidates (bv:334-355) */
    for (int i = 0; i < 32; i += 2) {
      const uint32_t i0 = surv[i / 2].iState << 1, i1 = i0 | 1u;
      const uint32_t os = surv[i / 2].oState << 2;
      cand[i].cost = surv[i / 2].cost; cand[i].oState = os | g[i0 & 0x1f]; cand[i].iState = i0;
      cand[i + 1].cost = surv[i / 2].cost; cand[i + 1].oState = os | g[i1 & 0x1f]; cand[i + 1].iState = i1;
    }
    /* getSoftCostMetrics (bv:358-368): cost += cTab[m&1][1] + cTab[(m>>1)&1][0] */
    const uint32_t in = history[ip];
    for (int i = 0; i < 32; i++) {
      const unsigned m = in ^ cand[i].oState;
      const float a = (m & 1u) ? mm[1] : mt[1];
      const float b = ((m >> 1) & 1u) ? mm[0] : mt[0];
      cand[i].cost += a + b;
    }
    /* pruneCandidates (bv:371-379) */
    for (int i = 0; i < 16; i++) {
      if (cand[i].cost < cand[i + 16].cost) { surv[i].iState = cand[i].iState; surv[i].oState = cand[i].oState; surv[i].cost = cand[i].cost; }
      else { surv[i].iState = cand[i + 16].iState; surv[i].oState = cand[i + 16].oState; surv[i].cost = cand[i + 16].cost; }
    }
    /* minCost (bv:382-393): first minimum */
    int mi = 0;
    float mc = surv[0].cost;
    for (int i = 1; i < 16; i++) {
      if (surv[i].cost >= mc) continue;
      mc = surv[i].cost; mi = i;
    }
    ip += 2; mt += 2; mm += 2;
    if (oc >= FO_DEFERRAL) out[op++] = (uint8_t)((surv[mi].iState >> FO_DEFERRAL) & 1u);
    oc++;
  }
}

/* Generator::encoderShift over the bits (bh:80-85, bv:208-214); state masked to psize bits */
uint64_t fo_parity(uint64_t coeff, unsigned psize, const uint8_t *bits, int n) {
  uint64_t st = 0;
  for (int i = 0; i < n; i++) {
    const unsigned fb = (unsigned)((st >> (psize - 1)) ^ bits[i]) & 1u;
    st <<= 1;
    if (fb) st ^= coeff;
  }
  return st & ((1ULL << psize) - 1);
}
/* Generator::syndromeShift over the bits (bh:69-74, bv:199-205) */
uint64_t fo_syndrome(uint64_t coeff, unsigned psize, const uint8_t *bits, int n) {
  uint64_t st = 0;
  for (int i = 0; i < n; i++) {
    const unsigned fb = (unsigned)(st >> (psize - 1)) & 1u;
    st = (st << 1) ^ (uint64_t)(bits[i] & 1u);
    if (fb) st ^= coeff;
  }
  return st & ((1ULL << psize) - 1);
}

/* BitVector::LSB8MSB (bv:166-195): reverse each whole octet */
void fo_lsb8msb(uint8_t *bits, int n) {
  for (int i = 0; i + 8 <= n; i += 8)
    for (int k = 0; k < 4; k++) { const uint8_t t = bits[i + k]; bits[i + k] = bits[i + 7 - k]; bits[i + 7 - k] = t; }
}

/* the soft value as the GSM side sees it after the UDP hop: (char)round(v*255.0) on the wire
   (Transceiver.cpp:669), byte/256.0F on arrival (TRXManager.cpp:231) */
float fo_wire(float v) {
  const int q = (int)round((double)v * 255.0);
  return (float)(unsigned char)q / 256.0F;
}

static uint64_t peek(const uint8_t *b, int at, int len) {   /* BitVector::peekField (bv:69-78) */
  uint64_t a = 0;
  for (int i = 0; i < len; i++) a = (a << 1) | (b[at + i] & 1u);
  return a;
}

/* XCCHL1Decoder::deinterleave + decode + mD.LSB8MSB() (fec:584-653): four bursts' e-bits i[B][114]
   (data1 = burst[3..60), data2 = burst[88..145), fec:607-608) -> u[228], d[184] */
int fo_xcch_decode(const float *i4x114, uint8_t *u228, uint8_t *d184, uint64_t *syn) {
  float c[456];
  for (int k = 0; k < 456; k++) {                           /* GSM 05.03 4.1.4 (fec:622-625) */
    const int B = k % 4, j = 2 * ((49 * k) % 57) + ((k % 8) / 4);
    c[k] = i4x114[B * 114 + j];
  }
  uint8_t u[228], dp[224];
  fo_viterbi_decode(c, 456, u, 228);
  if (u228) memcpy(u228, u, 228);
  memcpy(dp, u, 224);
  for (int i = 184; i < 224; i++) dp[i] ^= 1u;              /* mP.invert() (fec:644) */
  const uint64_t s = fo_syndrome(0x10004820009ULL, 40, dp, 224);
  if (syn) *syn = s;
  memcpy(d184, u, 184);
  fo_lsb8msb(d184, 184);
  return s == 0;
}

/* XCCHL1Encoder::sendFrame/encode/interleave/transmit (fec:772-845): 23 octets -> four 148-bit bursts (zero tails,
   stealing flags set fec:713-717, training sequence at 61, e-bits at 3..59 / 88..144) */
void fo_xcch_encode(const uint8_t *frame23, const uint8_t *tsc26, uint8_t *bursts4x148) {
  uint8_t u[228], c[456];
  memset(u, 0, sizeof u);
  for (int i = 0; i < 184; i++) u[i] = (frame23[i / 8] >> (7 - i % 8)) & 1u;   /* the L2 frame, MSB first */
  fo_lsb8msb(u, 184);                                                           /* fec:789 */
  const uint64_t pw = ~fo_parity(0x10004820009ULL, 40, u, 184);                 /* writeParityWord, inverted (bv:409-416) */
  for (int k = 0; k < 40; k++) u[184 + k] = (uint8_t)((pw >> (39 - k)) & 1u);
  fo_encode(u, 228, c);
  memset(bursts4x148, 0, 4 * 148);
  for (int B = 0; B < 4; B++) {
    bursts4x148[B * 148 + 60] = 1; bursts4x148[B * 148 + 87] = 1;
    for (int k = 0; k < 26; k++) bursts4x148[B * 148 + 61 + k] = tsc26[k] & 1u;
  }
  for (int k = 0; k < 456; k++) {
    const int B = k % 4, j = 2 * ((49 * k) % 57) + ((k % 8) / 4);
    bursts4x148[B * 148 + (j < 57 ? 3 + j : 88 + (j - 57))] = c[k];
  }
}

/* RACHL1Decoder::writeLowSide (fec:475-514) */
int fo_rach_decode(const float *e36, uint8_t *u18, unsigned *bsic, unsigned *ra) {
  uint8_t u[18], d[8];
  fo_viterbi_decode(e36, 36, u, 18);
  if (u18) memcpy(u18, u, 18);
  const int tail_ok = peek(u, 14, 4) == 0;
  const unsigned sent = ~(unsigned)peek(u, 8, 6);
  const unsigned chk = (unsigned)fo_parity(0x06f, 6, u, 8);
  if (bsic) *bsic = (sent ^ chk) & 0x3f;
  memcpy(d, u, 8);
  fo_lsb8msb(d, 8);
  if (ra) *ra = (unsigned)peek(d, 0, 8);
  return tail_ok;
}

/* TCHFACCHL1Decoder::decodeTCH(stolen = false) up to `good` (fec:1133-1163): class 1 = c[0..378) through the
   Viterbi decoder, class 2 = c[378..456) sliced; d[] reassembled; 3-bit parity (0x0b) over class 1a; tail */
int fo_tch_decode(const float *c456, uint8_t *u189, uint8_t *d260) {
  uint8_t u[189], d[260];
  fo_viterbi_decode(c456, 378, u, 189);
  for (int i = 0; i < 78; i++) d[182 + i] = c456[378 + i] > 0.5F;
  for (int k = 0; k <= 90; k++) { d[2 * k] = u[k]; d[2 * k + 1] = u[184 - k]; }
  const unsigned sent = (~(unsigned)peek(u, 91, 3)) & 7u;
  const unsigned calc = (unsigned)fo_parity(0x0b, 3, d, 50) & 7u;
  const unsigned tail = (unsigned)peek(u, 185, 4);
  if (u189) memcpy(u189, u, 189);
  memcpy(d260, d, 260);
  return sent == calc && tail == 0;
}

/* ---- batch forms over the burst path's output layout: soft[b][stride], 148 soft bits per burst ---- */

/* nblk blocks of four consecutive bursts.  wire != 0: the UDP quantisation in between.  frames: 23
   octets per block (d[] packed MSB first, BitVector::pack bv:541-552); ok[blk] = syndrome == 0 */
void fo_xcch_decode_batch(const float *soft, int stride, int nblk, int wire, uint8_t *frames, uint8_t *ok, int nthreads) {
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int blk = 0; blk < nblk; blk++) {
    float i4[4 * 114];
    for (int B = 0; B < 4; B++) {
      const float *s = soft + (size_t)(4 * blk + B) * stride;
      for (int k = 0; k < 57; k++) {
        i4[B * 114 + k] = wire ? fo_wire(s[3 + k]) : s[3 + k];
        i4[B * 114 + 57 + k] = wire ? fo_wire(s[88 + k]) : s[88 + k];
      }
    }
    uint8_t d[184];
    ok[blk] = (uint8_t)fo_xcch_decode(i4, NULL, d, NULL);
    for (int o = 0; o < 23; o++) frames[blk * 23 + o] = (uint8_t)peek(d, 8 * o, 8);
  }
}

/* one access burst per block: e = burst[49..85) (fec:479).  out[blk] = {tail_ok, bsic, ra} */
void fo_rach_decode_batch(const float *soft, int stride, int n, int wire, uint8_t *out3, int nthreads) {
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int b = 0; b < n; b++) {
    float e[36];
    for (int k = 0; k < 36; k++) e[k] = wire ? fo_wire(soft[(size_t)b * stride + 49 + k]) : soft[(size_t)b * stride + 49 + k];
    unsigned bsic, ra;
    out3[3 * b] = (uint8_t)fo_rach_decode(e, NULL, &bsic, &ra);
    out3[3 * b + 1] = (uint8_t)bsic;
    out3[3 * b + 2] = (uint8_t)ra;
  }
}

/* TCH/FACCH (fec:1030-1120): block m = bursts 4m .. 4m+7 of a traffic channel in arrival order, diagonal
   deinterleaver c[k] = i[(k + blockOffset) % 8][j(k)] (fec:1108-1116), which for consecutive bursts is burst
   4m + k%8.  nblk = nbursts/4 - 1.  Per block: tch[33] = d[260] packed MSB first (GSM 05.03 order, i.e. before
   the g610BitOrder unmap), tch_good; facch[23] + facch_ok = XCCH decode of the same c[] (what the reference
   runs when the frame is stolen); stolen = Hl (soft bit 60, fec:1077) of the block's last burst */
void fo_tch_decode_batch(const float *soft, int stride, int nbursts, int wire, uint8_t *tch, uint8_t *tch_good,
                         uint8_t *facch, uint8_t *facch_ok, uint8_t *stolen, int nthreads) {
  const int nblk = nbursts / 4 - 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int m = 0; m < nblk; m++) {
    float c[456];
    for (int k = 0; k < 456; k++) {
      const int j = 2 * ((49 * k) % 57) + ((k % 8) / 4);
      const float v = soft[(size_t)(4 * m + k % 8) * stride + (j < 57 ? 3 + j : 88 + (j - 57))];
      c[k] = wire ? fo_wire(v) : v;
    }
    uint8_t d[260], u[228], dd[184];
    tch_good[m] = (uint8_t)fo_tch_decode(c, NULL, d);
    for (int o = 0; o < 33; o++) {
      unsigned b = 0;
      for (int q = 0; q < 8; q++) b = (b << 1) | (8 * o + q < 260 ? d[8 * o + q] : 0u);
      tch[m * 33 + o] = (uint8_t)b;
    }
    /* XCCHL1Decoder::decode on the same c[] (fec:1079-1089) */
    fo_viterbi_decode(c, 456, u, 228);
    uint8_t dp[224];
    memcpy(dp, u, 224);
    for (int i = 184; i < 224; i++) dp[i] ^= 1u;
    facch_ok[m] = fo_syndrome(0x10004820009ULL, 40, dp, 224) == 0;
    memcpy(dd, u, 184);
    fo_lsb8msb(dd, 184);
    for (int o = 0; o < 23; o++) facch[m * 23 + o] = (uint8_t)peek(dd, 8 * o, 8);
    const float hl = soft[(size_t)(4 * m + 7) * stride + 60];
    stolen[m] = (wire ? fo_wire(hl) : hl) > 0.5F;
  }
}
