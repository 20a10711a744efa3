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

}  // extern "C"
