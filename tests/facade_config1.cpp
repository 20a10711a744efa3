// BASELINE config 1 through the product: the call sequence of the reference's Transceiver/sigProcLibTest.cpp:29-181 (sps 1,
// TSC 0; CommSig -> BitVector, SoftSig -> SoftVector as SURVEY section 4 describes) written against the reference's own
// function names and run through include/sigProcLib_trx.h, i.e. on the GPU -- every call of the reference's main(), in its
// order: :118 correlate, :122 vectorNorm2, :141-144 noisePwr / gaussianNoise (after srand(1), the C library's default
// seed, so the draws are the ones the reference's program makes) and :147 addVector included.  Inputs come from <dir>/in.bin
// (written by tests/test_facade.py from tests/golden/: the two bit patterns and the raw LPF tables -- reference data),
// every intermediate goes to <dir>/<name>.bin as raw float32 for the test to compare with tests/golden/config1_loopback.npz.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "sigProcLib_trx.h"

static std::string dir;
static void dump(const char *name, const float *p, size_t n) {
  FILE *f = std::fopen((dir + "/" + name + ".bin").c_str(), "wb");
  if (!f || std::fwrite(p, 4, n, f) != n) { std::fprintf(stderr, "cannot write %s\n", name); std::exit(20); }
  std::fclose(f);
}
static void dump(const char *name, const signalVector &v) { dump(name, (const float *)v.begin(), 2 * v.size()); }

int main(int argc, char **argv) {
  if (argc < 2) return 19;
  dir = argv[1];
  char bits[149] = {0}, rbits[149] = {0};
  std::vector<float> raw651(651), raw961(961);
  {
    FILE *f = std::fopen((dir + "/in.bin").c_str(), "rb");
    if (!f) return 18;
    bool ok = std::fread(bits, 1, 148, f) == 148 && std::fread(rbits, 1, 148, f) == 148 &&
              std::fread(raw651.data(), 4, 651, f) == 651 && std::fread(raw961.data(), 4, 961, f) == 961;
    std::fclose(f);
    if (!ok) return 17;
  }
  const int sps = 1;
  sigProcLibSetup(sps);
  if (!sigProcLibReady()) { std::fprintf(stderr, "sigProcLibSetup failed (no gfx950 device?)\n"); return 2; }
  signalVector *gsmPulse = generateGSMPulse(2, sps);
  if (!gsmPulse || !generateRACHSequence(*gsmPulse, sps) || !generateMidamble(*gsmPulse, sps, 0)) return 3;

  // ---- access burst (sigProcLibTest.cpp:38-53): modulate with guard 9, detectRACHBurst(.., 5.0, ..)
  BitVector RACHBurst(148);
  for (int k = 0; k < 148; k++) RACHBurst[k] = rbits[k];
  signalVector *rmod = modulateBurst(RACHBurst, *gsmPulse, 9, sps);
  if (!rmod) return 4;
  complex ramp; float rtoa = 0;
  const bool rfound = detectRACHBurst(*rmod, 5.0, sps, &ramp, &rtoa);
  dump("rach_x", *rmod);
  { const float r[4] = {rfound ? 1.0f : 0.0f, ramp.r, ramp.i, rtoa}; dump("rach", r, 4); }

  // ---- normal burst (:76-167)
  BitVector normalBurst(148);
  for (int k = 0; k < 148; k++) normalBurst[k] = bits[k];
  signalVector *mod = modulateBurst(normalBurst, *gsmPulse, 0, sps);                     // :84-85
  if (!mod) return 5;
  dump("mod", *mod);
  setLPFTables(raw651.data(), raw961.data());
  signalVector *lpfTx = createLPF(0.0f, 651, 96.0f);                                     // :91
  signalVector *lpfRx = createLPF(0.0f, 961, 65.0f);                                     // :98 (tap 960 of the raw table is 0)
  if (!lpfTx || !lpfRx) return 6;
  { std::vector<float> t(651); for (int k = 0; k < 651; k++) t[k] = (*lpfTx)[k].r; dump("lpf_tx", t.data(), 651); }
  { std::vector<float> t(961); for (int k = 0; k < 961; k++) t[k] = (*lpfRx)[k].r; dump("lpf_rx", t.data(), 961); }
  signalVector *up = polyphaseResampleVector(*mod, 96, 65, lpfTx);                       // :105-108
  if (!up) return 7;
  dump("up", *up);
  signalVector *dn = polyphaseResampleVector(*up, 65, 96, lpfRx);                        // :112-113
  if (!dn) return 8;
  dump("dn", *dn);
  signalVector *autocorr = correlate(dn, rmod, NULL, NO_DELAY);                          // :118
  if (!autocorr) return 14;
  dump("autocorr", *autocorr);
  { const float e = vectorNorm2(*up); dump("energy", &e, 1); }                           // :122
  delayVector(*dn, 6.932);                                                               // :125
  dump("delayed", *dn);
  signalVector channelResponse(4);                                                       // :133-137
  channelResponse[0] = complex(9000.0f, 0.0f);
  channelResponse[1] = complex(0.4f * 9000.0f, 0.0f);
  channelResponse[2] = complex(0.0f, 0.0f);
  channelResponse[3] = complex(-1.2f * 0.0f, 0.0f);
  signalVector *rx = convolve(dn, &channelResponse, NULL, NO_DELAY);                      // :139
  if (!rx) return 9;
  dump("rx", *rx);
  complex amp; float TOA = 0, chanOffset = 0;
  signalVector *chanResp = NULL;
  double noisePwr = 0.001 / sqrtf(2);                                                    // :143
  srand(1);                                                                              // (the seed a C program starts with)
  signalVector *noise = gaussianNoise((int)rx->size(), noisePwr);                        // :144
  if (!noise) return 15;
  dump("noise", *noise);
  const bool found = analyzeTrafficBurst(*rx, 0, 8.0, sps, &amp, &TOA, true, &chanResp, &chanOffset);   // :146
  { const float r[5] = {found ? 1.0f : 0.0f, amp.r, amp.i, TOA, chanOffset}; dump("det", r, 5); }
  if (!found || !chanResp) return 10;
  dump("chan", *chanResp);
  if (!addVector(*rx, *noise)) return 16;                                                // :147
  dump("rx_noisy", *rx);
  signalVector &rxNoisy = *rx;
  const float snr = 1.0 / noisePwr;                                                      // :159 (designDFE's float parameter)
  SoftVector *soft = demodulateBurst(rxNoisy, *gsmPulse, sps, amp, TOA);                 // :152
  if (!soft) return 11;
  dump("soft", soft->begin(), soft->size());
  signalVector *w = NULL, *b = NULL;
  if (!designDFE(*chanResp, snr, 7, &w, &b)) return 12;                                  // :159
  dump("dfe_w", *w); dump("dfe_b", *b);
  SoftVector *eq = equalizeBurst(rxNoisy, TOA - chanOffset, sps, *w, *b);                // :164
  if (!eq) return 13;
  dump("eq_soft", eq->begin(), eq->size());
  int errs = 0, eqerrs = 0;
  for (int k = 0; k < 148; k++) { errs += soft->bit(k) != normalBurst.bit(k); eqerrs += eq->bit(k) != normalBurst.bit(k); }
  std::printf("config 1: RACH found %d; TSC found %d, TOA %.4f; slicer bit errors %d, DFE bit errors %d\n", rfound, found, TOA, errs, eqerrs);
  delete eq; delete w; delete b; delete soft; delete chanResp; delete rx; delete dn; delete up; delete lpfTx; delete lpfRx;
  delete autocorr; delete noise;
  delete mod; delete rmod; delete gsmPulse;
  sigProcLibDestroy();
  return rfound ? 0 : 1;       // (the bit-error counts are the reference's own: the test compares them with the golden run)
}
