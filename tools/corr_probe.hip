// corr_probe.hip -- where does k_tsc_corr's time go?  A stand-alone harness around the product's own
// corr_issue / corr_round (csrc/trxsig_corr.h): the same kernel body with s_memrealtime (100 MHz) stamps per wave,
// random samples, K back-to-back launches; prints the phase durations and how many waves of the chip sit in
// which phase over time.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Iopenbts-ttsou_amd/csrc -Iinclude \
//         tools/corr_probe.hip openbts-ttsou_amd/csrc/trxsig_tablegen.cpp -o tools/corr_probe.bin && tools/corr_probe.bin
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "trxsig_corr.h"
#include "trxsig_tablegen.h"

#ifndef PROBE_WG
#define PROBE_WG 256
#endif
#ifndef PROBE_ROUNDS
#define PROBE_ROUNDS 1
#endif

namespace {
constexpr int SPS = 4;
constexpr int NST = 6;

template <unsigned TAPCLS, bool STAMPS>
__global__ __launch_bounds__(PROBE_WG) void k_probe(const cx *__restrict__ samples, const int32_t *__restrict__ offset,
                                                    const int32_t *__restrict__ length, int B, TapArg taps,
                                                    cx *__restrict__ rec, int Bpad, unsigned long long *__restrict__ dbg) {
  typedef CorrGeom<SPS> G;
  __shared__ __attribute__((aligned(16))) cx rows[PROBE_WG / 16][G::WPAD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = lane >> 4, r = lane & 15;
  const int slot = wave * 4 + row;
  unsigned long long st[NST];
  if (STAMPS) st[0] = wall_clock64();
  cx tap[16];
#pragma unroll
  for (int k = 0; k < 16; k++) tap[k] = mk(taps.v[2 * k], taps.v[2 * k + 1]);
#ifdef PROBE_STAGGER
  // experiment: the first generation of workgroups (the ones that start together on an empty chip) issue their loads
  // in six groups PROBE_STAGGER*64 cycles apart instead of all at once
  if (blockIdx.x < 1536) {
    const int g = blockIdx.x >> 8;
    for (int i = 0; i < g; i++) __builtin_amdgcn_s_sleep(PROBE_STAGGER);
  }
#endif
#ifdef PROBE_PIPE
  // experiment: PROBE_PIPE items per workgroup in sequence, the next item's loads issued before the current item's
  // arithmetic (a software pipeline), so that one generation of workgroups covers the whole batch
  {
    CorrIn<SPS> cur;
    corr_issue<SPS>(cur, (blockIdx.x * PROBE_PIPE + 0) * (PROBE_WG / 16) + slot, B, r, samples, offset, length);
    if (STAMPS) { st[1] = wall_clock64(); st[2] = st[1]; }
#pragma unroll 1
    for (int k = 0; k < PROBE_PIPE; k++) {
      CorrIn<SPS> nxt;
      const int kn = k + 1 < PROBE_PIPE ? k + 1 : k;         // (the last iteration re-loads its own item: in range, unused)
      corr_issue<SPS>(nxt, (blockIdx.x * PROBE_PIPE + kn) * (PROBE_WG / 16) + slot, B, r, samples, offset, length);
      __builtin_amdgcn_sched_barrier(0);
      int M;
      float energy;
      corr_round<SPS, true, true, TAPCLS>(cur, rows[slot], reinterpret_cast<float4 *>(rows[slot]), lane, r, tap, rec, Bpad, M, energy);
      __builtin_amdgcn_sched_barrier(0);
      cur = nxt;
    }
    if (STAMPS) {
      st[3] = wall_clock64();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      st[4] = wall_clock64();
      st[5] = 0;
      if (lane == 0) {
        unsigned long long *d = dbg + (size_t)(blockIdx.x * (PROBE_WG / 64) + wave) * NST;
        for (int k = 0; k < NST; k++) d[k] = st[k];
      }
    }
    return;
  }
#endif
  CorrIn<SPS> in[PROBE_ROUNDS];
#pragma unroll
  for (int i = 0; i < PROBE_ROUNDS; i++)
    corr_issue<SPS>(in[i], (blockIdx.x * PROBE_ROUNDS + i) * (PROBE_WG / 16) + slot, B, r, samples, offset, length);
  if (STAMPS) {
    st[1] = wall_clock64();
    // wait for the loads here so that the stamp separates "memory" from "math" (the product kernel waits at
    // first use, which is the same place)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st[2] = wall_clock64();
  }
#pragma unroll
  for (int i = 0; i < PROBE_ROUNDS; i++) {
    int M;
    float energy;
    corr_round<SPS, true, true, TAPCLS>(in[i], rows[slot], reinterpret_cast<float4 *>(rows[slot]), lane, r, tap, rec, Bpad, M, energy);
  }
  if (STAMPS) {
    st[3] = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st[4] = wall_clock64();
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    st[5] = ((unsigned long long)xcc << 32) | hw;
    if (lane == 0) {
      unsigned long long *d = dbg + (size_t)(blockIdx.x * (PROBE_WG / 64) + wave) * NST;
      for (int k = 0; k < NST; k++) d[k] = st[k];
    }
  }
}
}  // namespace

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
  const int B = 65536, K = argc > 1 ? std::atoi(argv[1]) : 200;
  static TrxTables T;
  if (trx_build_tables(&T, SPS) != 0) return 2;
  const int tsc = 2;
  TapArg ta;
  for (int k = 0; k < 16; k++) { ta.v[2 * k] = T.mid_ctap[tsc][k].r; ta.v[2 * k + 1] = T.mid_ctap[tsc][k].i; }
  if (tap_classes(&T, tsc) != TapPattern<SPS>::value) { std::fprintf(stderr, "unexpected tap classes\n"); return 3; }
  std::vector<int32_t> off(B), len(B);
  size_t tot = 0;
  for (int b = 0; b < B; b++) { len[b] = (b % 4 == 0) ? 628 : 624; off[b] = (int32_t)tot; tot += len[b]; }
  std::vector<cx> x(tot);
  unsigned s = 12345;
  for (auto &v : x) { s = s * 1664525u + 1013904223u; v.r = (float)((int)(s >> 9) % 2001 - 1000); s = s * 1664525u + 1013904223u; v.i = (float)((int)(s >> 9) % 2001 - 1000); }
  cx *dx, *drec; int32_t *doff, *dlen; unsigned long long *ddbg;
  const int Bpad = B, NS1 = CorrGeom<SPS>::NS + 1;
#ifdef PROBE_PIPE
  const int nwaves = B / 4 / PROBE_PIPE;
#else
  const int nwaves = B / 4;
#endif
  CK(hipMalloc(&dx, tot * sizeof(cx))); CK(hipMalloc(&drec, (size_t)NS1 * Bpad * sizeof(cx)));
  CK(hipMalloc(&doff, B * 4)); CK(hipMalloc(&dlen, B * 4)); CK(hipMalloc(&ddbg, (size_t)nwaves * NST * 8));
  CK(hipMemcpy(dx, x.data(), tot * sizeof(cx), hipMemcpyHostToDevice));
  CK(hipMemcpy(doff, off.data(), B * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dlen, len.data(), B * 4, hipMemcpyHostToDevice));
#ifdef PROBE_PIPE
  const dim3 grid(B / ((PROBE_WG / 16) * PROBE_PIPE)), block(PROBE_WG);
#else
  const dim3 grid(B / ((PROBE_WG / 16) * PROBE_ROUNDS)), block(PROBE_WG);
#endif
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < K; i++) k_probe<TapPattern<SPS>::value, false><<<grid, block>>>(dx, doff, dlen, B, ta, drec, Bpad, ddbg);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < K; i++) k_probe<TapPattern<SPS>::value, false><<<grid, block>>>(dx, doff, dlen, B, ta, drec, Bpad, ddbg);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::printf("WG %d threads, %d round(s): %.2f us per launch without stamps (back to back, %d launches)\n", PROBE_WG, PROBE_ROUNDS, ms * 1000 / K, K);
  CK(hipEventRecord(e0));
  for (int i = 0; i < K; i++) k_probe<TapPattern<SPS>::value, true><<<grid, block>>>(dx, doff, dlen, B, ta, drec, Bpad, ddbg);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::printf("with stamps: %.2f us per launch\n", ms * 1000 / K);
  std::vector<unsigned long long> d((size_t)nwaves * NST);
  CK(hipMemcpy(d.data(), ddbg, d.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long t0 = ~0ull, t1 = 0;
  for (int w = 0; w < nwaves; w++) { t0 = std::min(t0, d[(size_t)w * NST]); t1 = std::max(t1, d[(size_t)w * NST + 4]); }
  std::printf("last launch: first wave start -> last wave end %.2f us (100 MHz ticks)\n", (t1 - t0) / 100.0);
  const char *nm[] = {"issue loads", "wait for loads", "stage + energy + correlate + argmax + record", "wait for stores"};
  for (int k = 0; k < 4; k++) {
    double sum = 0; std::vector<double> v(nwaves);
    for (int w = 0; w < nwaves; w++) { v[w] = (d[(size_t)w * NST + k + 1] - d[(size_t)w * NST + k]) / 100.0; sum += v[w]; }
    std::sort(v.begin(), v.end());
    std::printf("  %-46s mean %6.2f us   p10 %6.2f  p50 %6.2f  p90 %6.2f\n", nm[k], sum / nwaves, v[nwaves / 10], v[nwaves / 2], v[nwaves * 9 / 10]);
  }
  // occupancy of the phases over time
  const int NB = 40; const double span = (double)(t1 - t0);
  std::vector<double> inLoad(NB, 0), inMath(NB, 0), inStore(NB, 0);
  for (int w = 0; w < nwaves; w++) {
    const unsigned long long *q = &d[(size_t)w * NST];
    auto add = [&](std::vector<double> &h, unsigned long long a, unsigned long long b) {
      for (int i = 0; i < NB; i++) {
        const double lo = t0 + span * i / NB, hi = t0 + span * (i + 1) / NB;
        const double o = std::min((double)b, hi) - std::max((double)a, lo);
        if (o > 0) h[i] += o / (hi - lo);
      }
    };
    add(inLoad, q[0], q[2]); add(inMath, q[2], q[3]); add(inStore, q[3], q[4]);
  }
  std::printf("  time(us)  waves loading / computing / draining stores   (chip holds %d wave slots at 6 per SIMD)\n", 256 * 4 * 6);
  for (int i = 0; i < NB; i++) std::printf("  %6.2f   %6.0f %6.0f %6.0f\n", span * (i + 0.5) / NB / 100.0, inLoad[i], inMath[i], inStore[i]);
  return 0;
}
