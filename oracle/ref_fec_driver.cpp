// oracle/ref_fec_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// extern "C" entry points around the UNMODIFIED reference CommonLibs/BitVector.{h,cpp} (compiled in
// place from /root/reference by `make -C oracle ref`, never copied): the rate-1/2 order-4 Viterbi
// coder (ViterbiR2O4), SoftVector::decode, BitVector::encode, the Parity/Generator shift registers and
// LSB8MSB.  The L1 FEC *flows* (GSM/GSML1FEC.cpp: RACHL1Decoder::writeLowSide :475-514,
// XCCHL1Decoder::deinterleave/decode :618-653, XCCHL1Encoder::encode/interleave :796-820) sit inside the
// GSM stack's threaded channel objects and cannot be built alone, so they are re-enacted here call by
// call on the reference's own primitives, in the reference's order; the interleaver index is the
// GSM 05.03 4.1.4 formula.  Used to pin oracle/fec_oracle.c and to generate tests/golden/fec_*.npz.
#include <stdint.h>
#include <string.h>

#include "BitVector.h"

extern "C" {

// SoftVector::decode (BitVector.cpp:438-524): n soft values -> nout bits
int reffec_soft_decode(const float *soft, int n, unsigned char *out, int nout) {
  SoftVector sv((size_t)n);
  for (int i = 0; i < n; i++) sv[i] = soft[i];
  BitVector target((size_t)nout);
  ViterbiR2O4 coder;
  sv.decode(coder, target);
  for (int i = 0; i < nout; i++) out[i] = target[i] & 0x01;
  return 0;
}

// BitVector::encode (BitVector.cpp:217-239): n bits -> 2n bits
int reffec_encode(const unsigned char *bits, int n, unsigned char *out) {
  BitVector src((size_t)n), dst((size_t)(2 * n));
  for (int i = 0; i < n; i++) src[i] = bits[i] & 0x01;
  ViterbiR2O4 coder;
  src.encode(coder, dst);
  for (int i = 0; i < 2 * n; i++) out[i] = dst[i] & 0x01;
  return 0;
}

uint64_t reffec_parity(uint64_t coeff, unsigned psize, unsigned cwsize, const unsigned char *bits, int n) {
  Parity p(coeff, psize, cwsize);
  BitVector b((size_t)n);
  for (int i = 0; i < n; i++) b[i] = bits[i];              // raw chars: consumers mask with 0x01
  return b.parity(p);
}

uint64_t reffec_syndrome(uint64_t coeff, unsigned psize, unsigned cwsize, const unsigned char *bits, int n) {
  Parity p(coeff, psize, cwsize);
  BitVector b((size_t)n);
  for (int i = 0; i < n; i++) b[i] = bits[i];
  return p.syndrome(b);
}

void reffec_lsb8msb(unsigned char *bits, int n) {
  BitVector b((size_t)n);
  for (int i = 0; i < n; i++) b[i] = bits[i];
  b.LSB8MSB();
  for (int i = 0; i < n; i++) bits[i] = b[i];
}

// XCCHL1Encoder::sendFrame/encode/interleave (GSML1FEC.cpp:772-820): d[184] (L2 bit order, before
// LSB8MSB) -> i[4][114] hard bits
void reffec_xcch_encode(const unsigned char *d184, unsigned char *i4x114) {
  Parity blockCoder(0x10004820009ULL, 40, 224);
  ViterbiR2O4 coder;
  BitVector mU(228), mC(456);
  mU.zero();
  BitVector mD(mU.head(184)), mP(mU.segment(184, 40));
  for (int i = 0; i < 184; i++) mD[i] = d184[i] & 0x01;
  mD.LSB8MSB();
  blockCoder.writeParityWord(mD, mP);
  mU.encode(coder, mC);
  for (int k = 0; k < 456; k++) {
    const int B = k % 4;
    const int j = 2 * ((49 * k) % 57) + ((k % 8) / 4);
    i4x114[B * 114 + j] = mC[k] & 0x01;
  }
}

// XCCHL1Decoder::deinterleave + decode (+ mD.LSB8MSB() of writeLowSide) (GSML1FEC.cpp:584-653):
// i[4][114] soft -> u[228] as decoded (parity field not yet inverted), d[184] after LSB8MSB; returns
// 1 when the syndrome is zero.  *syn receives the syndrome.
int reffec_xcch_decode(const float *i4x114, unsigned char *u228, unsigned char *d184, uint64_t *syn) {
  Parity blockCoder(0x10004820009ULL, 40, 224);
  ViterbiR2O4 coder;
  SoftVector mC(456);
  BitVector mU(228);
  BitVector mP(mU.segment(184, 40)), mDP(mU.head(224)), mD(mU.head(184));
  for (int k = 0; k < 456; k++) {
    const int B = k % 4;
    const int j = 2 * ((49 * k) % 57) + ((k % 8) / 4);
    mC[k] = i4x114[B * 114 + j];
  }
  mC.decode(coder, mU);
  for (int i = 0; i < 228; i++) u228[i] = mU[i] & 0x01;
  mP.invert();
  const uint64_t s = blockCoder.syndrome(mDP);
  if (syn) *syn = s;
  mD.LSB8MSB();
  for (int i = 0; i < 184; i++) d184[i] = mD[i] & 0x01;
  return s == 0;
}

// RACHL1Decoder::writeLowSide (GSML1FEC.cpp:475-514) up to the BSIC comparison: e[36] soft ->
// u[18]; returns 0 = tail bits non-zero, else 1; *bsic = the BSIC the parity word encodes,
// *ra = the RA byte (after LSB8MSB)
int reffec_rach_decode(const float *e36, unsigned char *u18, unsigned *bsic, unsigned *ra) {
  Parity parity(0x06f, 6, 8);
  ViterbiR2O4 coder;
  SoftVector e(36);
  for (int i = 0; i < 36; i++) e[i] = e36[i];
  BitVector mU(18);
  BitVector mD(mU.head(8));
  e.decode(coder, mU);
  for (int i = 0; i < 18; i++) u18[i] = mU[i] & 0x01;
  const int tail_ok = mU.peekField(14, 4) == 0;
  const unsigned sentParity = ~mU.peekField(8, 6);
  const unsigned checkParity = mD.parity(parity);
  if (bsic) *bsic = (sentParity ^ checkParity) & 0x03f;
  mD.LSB8MSB();
  if (ra) *ra = (unsigned)mD.peekField(0, 8);
  return tail_ok;
}

// TCHFACCHL1Encoder::encodeTCH (GSML1FEC.cpp:1251-1284) from d[260] in GSM 05.03 order (i.e. after the
// g610BitOrder map, which is a fixed permutation left to the caller): -> c[456]
void reffec_tch_encode(const unsigned char *d260, unsigned char *c456) {
  Parity tchParity(0x0b, 3, 50);
  ViterbiR2O4 coder;
  BitVector mC(456), mTCHU(189), mTCHD(260);
  BitVector mClass1_c(mC.head(378)), mClass1A_d(mTCHD.head(50)), mClass2_d(mTCHD.segment(182, 78));
  for (int i = 0; i < 260; i++) mTCHD[i] = d260[i] & 0x01;
  BitVector p = mTCHU.segment(91, 3);
  tchParity.writeParityWord(mClass1A_d, p);
  for (unsigned k = 0; k <= 90; k++) {
    mTCHU[k] = mTCHD[2 * k];
    mTCHU[184 - k] = mTCHD[2 * k + 1];
  }
  for (unsigned k = 185; k <= 188; k++) mTCHU[k] = 0;
  mTCHU.encode(coder, mClass1_c);
  mClass2_d.copyToSegment(mC, 378);
  for (int i = 0; i < 456; i++) c456[i] = mC[i] & 0x01;
}

// TCHFACCHL1Decoder::decodeTCH(stolen = false) up to `good` (GSML1FEC.cpp:1133-1163): c[456] soft ->
// u[189], d[260]; returns good
int reffec_tch_decode(const float *c456, unsigned char *u189, unsigned char *d260) {
  Parity tchParity(0x0b, 3, 50);
  ViterbiR2O4 coder;
  SoftVector mC(456);
  for (int i = 0; i < 456; i++) mC[i] = c456[i];
  BitVector mTCHU(189), mTCHD(260);
  SoftVector mClass1_c(mC.head(378)), mClass2_c(mC.segment(378, 78));
  BitVector mClass1A_d(mTCHD.head(50));
  mClass1_c.decode(coder, mTCHU);
  mClass2_c.sliced().copyToSegment(mTCHD, 182);
  for (unsigned k = 0; k <= 90; k++) {
    mTCHD[2 * k] = mTCHU[k];
    mTCHD[2 * k + 1] = mTCHU[184 - k];
  }
  const unsigned sentParity = (~mTCHU.peekField(91, 3)) & 0x07;
  const unsigned calcParity = mClass1A_d.parity(tchParity) & 0x07;
  const unsigned tail = mTCHU.peekField(185, 4);
  for (int i = 0; i < 189; i++) u189[i] = mTCHU[i] & 0x01;
  for (int i = 0; i < 260; i++) d260[i] = mTCHD[i] & 0x01;
  return (sentParity == calcParity) && (tail == 0);
}

}  // extern "C"
