// tools/group_tx_bench.cpp -- tools/group_tx_bench.py's step as a C++ host would drive it (the reference's own host language), straight
// on the C-ABI: per step F frames of S ARFCNs (every timeslot a burst) are "received" into the group's pinned staging block -- the
// payloads are there already, only the frame numbers are written, as a recvfrom() would leave them --, added
// (trxsig_trxgroup_add_staged), pushed into the fused transmit back end (trxsig_trxgroup_push_txbe) and popped (trxsig_txbe_pop).
//   make -C openbts-ttsou_amd/csrc tx_bench && openbts-ttsou_amd/tx_bench [S] [frames per step] [steps]
// One JSON line: the loop's wall time per step (a device synchronise behind the last step), and the host's time in each call.
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "trxsig.h"
#include "trxsig_frontend.h"
#include "trxsig_trxgroup.h"

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define OK(x) do { const int rc_ = (x); if (rc_ != 0) { std::fprintf(stderr, "%s: %d\n", #x, rc_); return 1; } } while (0)

int main(int argc, char **argv) {
  const int S = argc > 1 ? std::atoi(argv[1]) : 128, F = argc > 2 ? std::atoi(argv[2]) : 8, K = argc > 3 ? std::atoi(argv[3]) : 200;
  const int sps = 4, n = S * 8 * F;
  trxsig_ctx *ctx = nullptr;
  OK(trxsig_create(&ctx, 0, sps));
  trxsig_trxgroup *grp = nullptr;
  OK(trxsig_trxgroup_create(&grp, ctx, S, TRXSIG_TSCLEG_DEMOD, 0, 0));
  char resp[64];
  for (int a = 0; a < S; a++)
    for (int tn = 0; tn < 8; tn++) {
      char cmd[64];
      std::snprintf(cmd, sizeof cmd, "CMD SETSLOT %d %d", tn, (tn == 0 && a % 8 == 0) ? 5 : 1);
      if (trxsig_trxgroup_control(grp, a, cmd, resp, sizeof resp) < 0) { std::fprintf(stderr, "SETSLOT refused\n"); return 1; }   // (returns the response's length)
    }
  // a 651-tap windowed-sinc low-pass of gain 96 (any taps do for a timing; tools/group_tx_bench.py uses the designed Kaiser filter)
  const int L = 651;
  std::vector<float> lpf((size_t)L);
  double sum = 0.0;
  for (int i = 0; i < L; i++) {
    const double x = (i - (L - 1) / 2.0) / 96.0, w = 0.5 - 0.5 * std::cos(2.0 * M_PI * i / (L - 1));
    const double v = (x == 0.0 ? 1.0 : std::sin(M_PI * x) / (M_PI * x)) * w;
    lpf[(size_t)i] = (float)v; sum += v;
  }
  for (float &v : lpf) v = (float)(v * 96.0 / sum);
  trxsig_txbe *be = nullptr;
  OK(trxsig_txbe_create(&be, ctx, S, 8 * F, lpf.data(), L, 1.0f));
  // the batch: every (ARFCN, frame offset, timeslot) once, arrival order shuffled
  std::mt19937 rng(3);
  std::vector<int> arf((size_t)n), fo((size_t)n), tn((size_t)n), perm((size_t)n);
  for (int i = 0; i < n; i++) { arf[i] = i / (8 * F); fo[i] = (i / 8) % F; tn[i] = i % 8; perm[i] = i; }
  for (int i = n - 1; i > 0; i--) std::swap(perm[i], perm[(size_t)(rng() % (unsigned)(i + 1))]);
  std::vector<uint8_t> base((size_t)n * 154);
  for (int i = 0; i < n; i++) {
    uint8_t *d = &base[(size_t)i * 154];
    d[0] = (uint8_t)tn[perm[i]]; d[5] = (uint8_t)(rng() % 30);
    for (int b = 0; b < 148; b++) d[6 + b] = (uint8_t)(rng() & 1);
  }
  int fn = 1000, filled[4] = {0, 0, 0, 0};
  uint8_t *seen[4] = {nullptr, nullptr, nullptr, nullptr};
  double t_stage = 0, t_recv = 0, t_add = 0, t_push = 0, t_pop = 0, t0 = 0;
  int n_out = 0;
  for (int it = 0; it < K + 10; it++) {
    if (it == 10) { OK((int)hipDeviceSynchronize()); t_stage = t_recv = t_add = t_push = t_pop = 0; t0 = now_us(); }
    const double a0 = now_us();
    uint8_t *d = nullptr; int32_t *ar = nullptr;
    OK(trxsig_trxgroup_tx_staging(grp, n, &d, &ar));
    const double a1 = now_us();
    int k = -1;
    for (int j = 0; j < 4; j++) if (seen[j] == d) k = j;
    if (k < 0) { k = 0; while (k < 3 && seen[k]) k++; seen[k] = d; filled[k] = 0; }
    if (!filled[k]) {                                       // (a few blocks take turns: the payloads are the same every step, written once per block)
      std::memcpy(d, base.data(), base.size());
      for (int i = 0; i < n; i++) ar[i] = arf[perm[i]];
      filled[k] = 1;
    }
    for (int i = 0; i < n; i++) {                           // this step's frame numbers, big-endian
      const uint32_t f = (uint32_t)(fn + fo[perm[i]]);
      uint8_t *h = d + (size_t)i * 154 + 1;
      h[0] = (uint8_t)(f >> 24); h[1] = (uint8_t)(f >> 16); h[2] = (uint8_t)(f >> 8); h[3] = (uint8_t)f;
    }
    const double a2 = now_us();
    OK(trxsig_trxgroup_add_staged(grp, n));
    const double a3 = now_us();
    OK(trxsig_trxgroup_push_txbe(grp, be, fn, 0, 8 * F));
    const double a4 = now_us();
    const int16_t *iq = nullptr; int64_t stride = 0;
    OK(trxsig_txbe_pop(be, &iq, &stride, &n_out));
    const double a5 = now_us();
    t_stage += a1 - a0; t_recv += a2 - a1; t_add += a3 - a2; t_push += a4 - a3; t_pop += a5 - a4;
    fn += F;
  }
  OK((int)hipDeviceSynchronize());
  const double dt = (now_us() - t0) / K;
  int dropped = 0;
  const int left = trxsig_trxgroup_tx_queue_size(grp, 0, &dropped);   // (the queue's size, or a negative code)
  std::printf("{\"host\": \"C++ on the C-ABI (tools/group_tx_bench.cpp)\", \"arfcns\": %d, \"frames_per_step\": %d, \"bursts_per_step\": %d, \"steps\": %d, "
              "\"us_per_step\": %.1f, \"Mbursts_per_s\": %.2f, \"host_us_writing_frame_numbers\": %.1f, \"host_us_in_tx_staging\": %.1f, "
              "\"host_us_in_add_staged\": %.1f, \"host_us_in_push_txbe\": %.1f, \"host_us_in_txbe_pop\": %.1f, \"int16_pairs_out_per_stream\": %d, "
              "\"queue_left\": %d, \"dropped\": %s}\n",
              S, F, n, K, dt, n / dt, t_recv / K, t_stage / K, t_add / K, t_push / K, t_pop / K, n_out, left, dropped ? "true" : "false");
  trxsig_txbe_destroy(be); trxsig_trxgroup_destroy(grp); trxsig_destroy(ctx);
  return 0;
}
