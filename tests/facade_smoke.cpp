// Loopback through the sigProcLib.h-compatible facade (include/sigProcLib_trx.h): the call sequence of
// the reference's Transceiver/sigProcLibTest.cpp (modulate -> analyzeTrafficBurst -> demodulateBurst, and
// the RACH leg) written against the reference's own function names.  Exit code 0 = all bits recovered.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "sigProcLib_trx.h"

int main(int argc, char **argv) {
  const int sps = argc > 1 ? std::atoi(argv[1]) : 4;
  sigProcLibSetup(sps);
  if (!sigProcLibReady()) { std::fprintf(stderr, "sigProcLibSetup failed (no gfx950 device?)\n"); return 2; }
  signalVector *gsmPulse = generateGSMPulse(2, sps);
  if (!gsmPulse) return 3;
  generateRACHSequence(*gsmPulse, sps);
  generateMidamble(*gsmPulse, sps, 0);

  // normal burst: sigProcLibTest.cpp:76-78 payload + TSC 0 + payload
  const char *seg = "0000101010100111110010101010010110101110011000111001101010000";
  const char *tsc0 = "00100101110000100010010111";
  char all[149];
  std::snprintf(all, sizeof all, "%s%s%s", seg, tsc0, seg);
  BitVector normalBurst(all);
  signalVector *mod = modulateBurst(normalBurst, *gsmPulse, 8, sps);
  if (!mod) return 4;
  for (size_t k = 0; k < mod->size(); k++) (*mod)[k] = complex((*mod)[k].r * 1000.0f, (*mod)[k].i * 1000.0f);

  float avgPwr = 0;
  const bool e = energyDetect(*mod, 20 * sps, 1.0f, &avgPwr);
  complex amp; float toa = 0;
  const bool found = analyzeTrafficBurst(*mod, 0, 3.0, sps, &amp, &toa);
  std::printf("energy %d (avgPwr %.1f), TSC found %d, amp (%.4f,%.4f), TOA %.4f\n", e, avgPwr, found, amp.r, amp.i, toa);
  if (!found || !e) return 5;
  SoftVector *soft = demodulateBurst(*mod, *gsmPulse, sps, amp, toa);
  if (!soft) return 6;
  int errs = 0;
  for (int k = 0; k < 148; k++) errs += soft->bit(k) != normalBurst.bit(k);
  std::printf("normal burst: %d soft bits, %d bit errors; first soft %.6f\n", (int)soft->size(), errs, (*soft)[0]);

  // access burst: sigProcLibTest.cpp:38-53
  char rb[149];
  std::snprintf(rb, sizeof rb, "01010101%s%099d", "01001011011111111001100110101010001111000", 0);
  BitVector rach(rb);
  signalVector *rmod = modulateBurst(rach, *gsmPulse, 9, sps);
  if (!rmod) return 7;
  for (size_t k = 0; k < rmod->size(); k++) (*rmod)[k] = complex((*rmod)[k].r * 500.0f, (*rmod)[k].i * 500.0f);
  complex ramp; float rtoa = 0;
  const bool rfound = detectRACHBurst(*rmod, 5.0, sps, &ramp, &rtoa);
  std::printf("RACH found %d, amp (%.3f,%.3f), TOA %.5f\n", rfound, ramp.r, ramp.i, rtoa);
  int rerrs = 0;
  if (rfound) {
    SoftVector *rs = demodulateBurst(*rmod, *gsmPulse, sps, ramp, rtoa);
    if (!rs) return 8;
    for (int k = 0; k < 148; k++) rerrs += rs->bit(k) != rach.bit(k);
    delete rs;
  }
  std::printf("access burst: %d bit errors\n", rerrs);

  // rate conversion as RadioInterface::pullBuffer does it (radioInterface.cpp:230-246): createLPF + polyphaseResampleVector
  std::vector<float> raw651(651), raw961(961);
  for (int k = 0; k < 651; k++) raw651[k] = 0.001f * (float)((k * 37) % 101) - 0.02f;
  for (int k = 0; k < 961; k++) raw961[k] = 0.001f * (float)((k * 53) % 97) - 0.01f;
  setLPFTables(raw651.data(), raw961.data());
  signalVector *lpf = createLPF(1.0f / 96.0f, 961, (float)(65 * sps));
  signalVector in(192 + 864);
  for (size_t k = 0; k < in.size(); k++) in[k] = complex((float)((k * 7) % 13) - 6.0f, (float)((k * 11) % 17) - 8.0f);
  signalVector *res = lpf ? polyphaseResampleVector(in, 65 * sps, 96, lpf) : NULL;
  std::printf("resampled %d -> %d samples\n", (int)in.size(), res ? (int)res->size() : -1);
  const bool rs_ok = res && (int)res->size() == (int)((in.size() * 65 * sps + 95) / 96);
  delete res; delete lpf;

  // the equaliser leg of Transceiver::pullRadioVector (Transceiver.cpp:317-349, 391-396), symbol-rate samples only:
  // analyzeTrafficBurst(requestChannel) -> scaleVector(chan, 1/amp) -> designDFE -> scaleVector(burst, 1/amp) -> equalizeBurst
  int eqerrs = 0;
  if (sps == 1) {
    const char *tsc6 = "10100111110110001010011111";
    char b6[149];
    std::snprintf(b6, sizeof b6, "%s%s%s", seg, tsc6, seg);
    BitVector burst6(b6);
    signalVector *m6 = modulateBurst(burst6, *gsmPulse, 8, sps);
    if (!m6) return 9;
    for (size_t k = 0; k < m6->size(); k++) (*m6)[k] = (*m6)[k] * complex(600.0f, 400.0f);
    for (size_t k = m6->size() - 1; k >= 1; k--) (*m6)[k] = complex((*m6)[k].r + 0.3f * (*m6)[k - 1].r, (*m6)[k].i + 0.3f * (*m6)[k - 1].i);   // a second path
    complex a6; float t6 = 0, choff = 0;
    signalVector *chan = NULL, *w = NULL, *b = NULL;
    const bool f6 = analyzeTrafficBurst(*m6, 6, 3.0, sps, &a6, &t6, true, &chan, &choff);
    if (!f6 || !chan) return 10;
    scaleVector(*chan, complex(1.0, 0.0) / a6);
    if (!designDFE(*chan, 100.0f, 7, &w, &b)) return 11;
    scaleVector(*m6, complex(1.0, 0.0) / a6);
    SoftVector *es = equalizeBurst(*m6, t6 - choff, sps, *w, *b);
    if (!es) return 12;
    for (int k = 0; k < 148; k++) eqerrs += es->bit(k) != burst6.bit(k);
    std::printf("equalised burst: channel offset %.1f, |w3| %.4f, %d bit errors\n", choff, (*w)[3].norm2(), eqerrs);
    delete es; delete w; delete b; delete chan; delete m6;
  }

  delete soft; delete mod; delete rmod; delete gsmPulse;
  sigProcLibDestroy();
  return (errs == 0 && rfound && rerrs == 0 && rs_ok && eqerrs == 0) ? 0 : 1;
}
