// trxsig_udp_main.cpp -- the `transceiver` process of the reference (Transceiver/runTransceiver.cpp + the four service loops of
// Transceiver.cpp:582-793) for N ARFCNs on ONE GPU: a socket loop around the Transceiver GROUP (include/trxsig_trxgroup.h) that
// speaks the TRXManager UDP contract (TRXManager/README.TRXManager) with the port plan TransceiverManager lays out
// (TRXManager/TRXManager.cpp:44-54, 123-124; Transceiver.cpp:45-47):
//     clock indications          port B            -> peer B+100          "IND CLOCK <fn>"        (one socket for the whole process)
//     ARFCN i control            port B+1+2i      <-> peer B+101+2i       "CMD ..." / "RSP ..."
//     ARFCN i data               port B+2+2i      <-> peer B+102+2i       154-byte transmit bursts in, 158-byte receive bursts out
// Per frame of the radio clock (8 timeslots x N ARFCNs) the loop does what the reference's threads do per burst:
//   driveControl                (:439-580)  every control socket, answered through trxsig_trxgroup_control; a clock indication per
//                                           command as in :463;
//   driveTransmitPriorityQueue  (:582-639)  every data socket drained into ONE trxsig_trxgroup_add_bursts; a datagram of the wrong
//                                           length is dropped and the core reminded of the clock after the next good one
//                                           (TransmitPriorityQueueServiceLoopAdapter :778-793); the periodic reminder :617-618;
//   driveTransmitFIFO           (:679-729)  trxsig_txclock_advance: the deadline clock runs mTransmitLatency ahead of the radio clock,
//                                           +1 frame after an under-run, -1 timeslot after 216 quiet frames; the due timeslots leave
//                                           through ONE trxsig_trxgroup_push (priority queue / stale dump / filler table on the device);
//   driveReceiveFIFO            (:641-677)  ONE trxsig_trxgroup_pull for the frame, trxsig_trxgroup_collect, a 158-byte datagram
//                                           per SoftVector that came back.
// There is no radio in this image: the samples take the software-loopback route of the reference's SWLOOPBACK build -- what
// pushRadioVector hands to the transmit FIFO for slot (fn, tn) is modulated on the device (trxsig_modulate_batch with the gain,
// x13500 as RadioInterface::pushBuffer scales to the radio's range, radioInterface.cpp:149) into a ring of frames and comes back
// as the receive burst of the same slot.  A real radio replaces `Loopback` with trxsig_txbe / trxsig_rxfe (trxsig_frontend.h).
// The radio clock is the wall clock (--frame-us per frame, 4615 = real time); a loop that falls more than the transmit latency
// behind it has under-run the radio, which is what drives the latency controller.  Runs until SIGINT / SIGTERM (or --frames).
//
//   trxsig_transceiver_udp [--port 5700] [--arfcns 1] [--sps 1] [--tsc-leg equalize|demod] [--frame-us 4615] [--frames 0]
//                          [--device 0] [--stall-frame K --stall-ms M]   (test hook: one late frame, i.e. one under-run)
#include <arpa/inet.h>
#include <hip/hip_runtime_api.h>
#include <netinet/in.h>
#include <signal.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "trxsig_trxgroup.h"

namespace {

constexpr int kHyperframe = 2048 * 26 * 51;
volatile sig_atomic_t g_stop = 0;
void on_signal(int) { g_stop = 1; }

int open_udp(int local_port) {
  const int fd = socket(AF_INET, SOCK_DGRAM, 0);
  if (fd < 0) return -1;
  sockaddr_in a{};
  a.sin_family = AF_INET;
  a.sin_port = htons((uint16_t)local_port);
  a.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
  if (bind(fd, (sockaddr *)&a, sizeof a) != 0) { close(fd); return -1; }
  int sz = 4 << 20;                                         // a frame of 128 ARFCNs is 1024 datagrams each way
  (void)setsockopt(fd, SOL_SOCKET, SO_RCVBUF, &sz, sizeof sz);
  (void)setsockopt(fd, SOL_SOCKET, SO_SNDBUF, &sz, sizeof sz);
  return fd;
}
void send_to(int fd, int port, const void *buf, size_t n) {
  sockaddr_in a{};
  a.sin_family = AF_INET;
  a.sin_port = htons((uint16_t)port);
  a.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
  (void)sendto(fd, buf, n, 0, (sockaddr *)&a, sizeof a);
}
long recv_nb(int fd, void *buf, size_t cap) { return recv(fd, buf, cap, MSG_DONTWAIT); }

#define CHK(call, what)                                                                                    \
  do {                                                                                                     \
    const int rc_ = (call);                                                                                \
    if (rc_ < 0) { std::fprintf(stderr, "%s: %d (%s)\n", what, rc_, trxsig_last_error(ctx)); return 4; }   \
  } while (0)

// The software-loopback radio: a ring of frames of modulated bursts, cell (frame % R, tn, arfcn) of `cell` complex samples.
struct Loopback {
  trxsig_ctx *ctx = nullptr;
  int S = 0, sps = 1, cell = 0, R = 64;
  trxsig_c32 *d_ring = nullptr;
  int32_t *d_meta = nullptr;                                // guard[cap] | off[cap] | len[cap]
  trxsig_c32 *d_scale = nullptr;                            // (13500, 0) x cap
  int cap = 0;
  int init(trxsig_ctx *c, int S_, int sps_) {
    ctx = c; S = S_; sps = sps_; cell = 160 * sps; cap = 64 * S;
    if (hipMalloc((void **)&d_ring, sizeof(trxsig_c32) * (size_t)R * 8 * S * cell) != hipSuccess) return -1;
    if (hipMemset(d_ring, 0, sizeof(trxsig_c32) * (size_t)R * 8 * S * cell) != hipSuccess) return -1;
    if (hipMalloc((void **)&d_meta, sizeof(int32_t) * 3 * (size_t)cap) != hipSuccess) return -1;
    if (hipMalloc((void **)&d_scale, sizeof(trxsig_c32) * (size_t)cap) != hipSuccess) return -1;
    std::vector<trxsig_c32> sc((size_t)cap, trxsig_c32{13500.0f, 0.0f});   // RadioInterface::pushBuffer's scaleVector (radioInterface.cpp:149)
    return hipMemcpy(d_scale, sc.data(), sizeof(trxsig_c32) * sc.size(), hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
  }
  size_t cell_index(int fn, int tn, int a) const { return ((((size_t)(fn % R) * 8) + tn) * S + a) * (size_t)cell; }
  // what trxsig_trxgroup_push handed out for n timeslots from (fn, tn) -- [S][n][148] bits, [S][n] gains -- onto the air
  int transmit(const uint8_t *d_bits, const float *d_gain, int fn, int tn, int n) {
    if (n * S > cap) return TRXSIG_EINVAL;
    std::vector<int32_t> meta(3 * (size_t)cap, 0);
    for (int a = 0; a < S; a++) {
      int f = fn, t = tn;
      for (int k = 0; k < n; k++) {
        const size_t i = (size_t)a * n + k;
        const int guard = 8 + ((t % 4) == 0);               // Transceiver.cpp:105
        meta[i] = guard;
        meta[(size_t)cap + i] = (int32_t)cell_index(f, t, a);
        meta[2 * (size_t)cap + i] = sps * (148 + guard);
        if (++t > 7) { t = 0; f = (f + 1) % kHyperframe; }
      }
    }
    hipStream_t st = (hipStream_t)trxsig_get_stream(ctx);
    if (hipMemcpyAsync(d_meta, meta.data(), sizeof(int32_t) * meta.size(), hipMemcpyHostToDevice, st) != hipSuccess) return TRXSIG_EHIP;
    if (hipStreamSynchronize(st) != hipSuccess) return TRXSIG_EHIP;   // (meta is a local)
    int rc = trxsig_modulate_batch(ctx, d_bits, d_meta, d_gain, n * S, d_ring, d_meta + cap);
    if (rc != TRXSIG_OK) return rc;
    return trxsig_scale_vector_batch(ctx, d_ring, d_meta + cap, d_meta + 2 * cap, n * S, sps * 157, d_scale, 0);
  }
  const trxsig_c32 *frame(int fn) const { return d_ring + cell_index(fn, 0, 0); }
};

}  // namespace

int main(int argc, char **argv) {
  int port = 5700, N = 1, sps = 1, device = 0, frame_us = 4615, frames = 0, stall_frame = -1, stall_ms = 0, dbg_arfcn = -1;
  const char *leg_name = nullptr;
  for (int i = 1; i + 1 < argc; i += 2) {
    if (!std::strcmp(argv[i], "--port")) port = std::atoi(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--arfcns")) N = std::atoi(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--sps")) sps = std::atoi(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--tsc-leg")) leg_name = argv[i + 1];
    else if (!std::strcmp(argv[i], "--frame-us")) frame_us = std::atoi(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--slot-us")) frame_us = 8 * std::atoi(argv[i + 1]);     // (the one-ARFCN harness's option)
    else if (!std::strcmp(argv[i], "--frames")) frames = std::atoi(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--device")) device = std::atoi(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--stall-frame")) stall_frame = std::atoi(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--stall-ms")) stall_ms = std::atoi(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--debug-arfcn")) dbg_arfcn = std::atoi(argv[i + 1]);
    else { std::fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
  }
  if (N < 1 || N > 1024 || frame_us < 1) { std::fprintf(stderr, "--arfcns 1..1024, --frame-us >= 1\n"); return 2; }
  const int leg = leg_name ? (!std::strcmp(leg_name, "demod") ? TRXSIG_TSCLEG_DEMOD : TRXSIG_TSCLEG_EQUALIZE)
                           : (sps == 1 ? TRXSIG_TSCLEG_EQUALIZE : TRXSIG_TSCLEG_DEMOD);
  signal(SIGINT, on_signal); signal(SIGTERM, on_signal);

  const int clk = open_udp(port);
  std::vector<int> ctl((size_t)N), dat((size_t)N);
  bool ok = clk >= 0;
  for (int i = 0; i < N && ok; i++) {
    ctl[(size_t)i] = open_udp(port + 1 + 2 * i);
    dat[(size_t)i] = open_udp(port + 2 + 2 * i);
    ok = ctl[(size_t)i] >= 0 && dat[(size_t)i] >= 0;
  }
  if (!ok) { std::fprintf(stderr, "cannot bind UDP ports %d..%d\n", port, port + 2 * N); return 2; }

  trxsig_ctx *ctx = nullptr;
  if (trxsig_create(&ctx, device, sps) != TRXSIG_OK) { std::fprintf(stderr, "no context (GPU?)\n"); return 3; }
  const int start_fn = 2;                                   // (the reference starts at a random frame number, Transceiver.cpp:50)
  trxsig_trxgroup *grp = nullptr;
  CHK(trxsig_trxgroup_create(&grp, ctx, N, leg, start_fn, 0), "trxsig_trxgroup_create");
  Loopback air;
  if (air.init(ctx, N, sps) != 0) { std::fprintf(stderr, "loopback ring: out of device memory\n"); return 3; }
  trxsig_txclock txc;
  trxsig_txclock_init(&txc, start_fn, 0, 2, 0);             // runTransceiver.cpp:53: transmit latency GSM::Time(2,0)

  auto write_clock = [&]() {                                // writeClockInterface (:733-746)
    char msg[64];
    const int n = trxsig_txclock_indication(&txc, msg, sizeof msg);
    if (n > 0) send_to(clk, port + 100, msg, (size_t)n + 1);
  };
  const size_t cells = (size_t)8 * N;
  std::vector<uint8_t> valid(cells), dgs;
  std::vector<int32_t> arf;
  std::vector<float> soft(cells * 148);
  std::vector<int> rssi(cells), timing(cells);
  long rx_sent = 0, tx_recv = 0, bad_len = 0, underruns = 0, clock_inds = 0;
  double busy_sum = 0, busy_max = 0;
  using clock_t_ = std::chrono::steady_clock;
  const auto t_start = clock_t_::now();
  write_clock(); clock_inds++;
  bool remind = false;                                      // a malformed datagram was flushed: remind the core of the clock (:785-789)
  int pending_underrun = 0;
  long k = 0;
  for (; !g_stop && (frames <= 0 || k < frames); k++) {
    const int fn = (int)((start_fn + k) % kHyperframe);     // the radio clock: frame fn is on the air
    const auto due = t_start + std::chrono::microseconds((long long)k * frame_us);
    const auto now0 = clock_t_::now();
    if (now0 < due) std::this_thread::sleep_until(due);
    else {
      // the radio has consumed (now - due) worth of samples this loop did not provide in time: more than the transmit latency
      // behind is an under-run at the device (radioInterface.cpp:165-168 reports it from writeSamples)
      const double late_frames = std::chrono::duration<double, std::micro>(now0 - due).count() / frame_us;
      if (late_frames > txc.latency_fn + txc.latency_tn / 8.0) { pending_underrun = 1; underruns++; }
    }
    if (k == stall_frame && stall_ms > 0) std::this_thread::sleep_for(std::chrono::milliseconds(stall_ms));   // test hook
    const auto t0 = clock_t_::now();

    // ---- driveControl ----
    for (int i = 0; i < N; i++) {
      char cbuf[128];
      for (long n; (n = recv_nb(ctl[(size_t)i], cbuf, sizeof cbuf - 1)) > 0;) {
        cbuf[n] = 0;
        char rsp[128];
        write_clock(); clock_inds++;                        // (:463: before the command is looked at)
        const int rn = trxsig_trxgroup_control(grp, i, cbuf, rsp, sizeof rsp);
        if (rn > 0) send_to(ctl[(size_t)i], port + 101 + 2 * i, rsp, (size_t)rn + 1);
      }
    }
    // ---- driveTransmitPriorityQueue ----
    dgs.clear(); arf.clear();
    for (int i = 0; i < N; i++) {
      uint8_t dbuf[256];
      for (long n; (n = recv_nb(dat[(size_t)i], dbuf, sizeof dbuf)) > 0;) {
        const int fn_dg = n >= 5 ? (int)(((uint32_t)dbuf[1] << 24) | ((uint32_t)dbuf[2] << 16) | ((uint32_t)dbuf[3] << 8) | dbuf[4]) : -1;
        if (n != TRXSIG_TX_DATAGRAM_BYTES || dbuf[0] > 7 || fn_dg < 0 || fn_dg >= kHyperframe) {   // "badly formatted packet" (:592-595)
          bad_len++; remind = true;
          continue;
        }
        if (trxsig_txclock_indication_due(&txc)) { write_clock(); clock_inds++; }                 // (:617-618)
        dgs.insert(dgs.end(), dbuf, dbuf + TRXSIG_TX_DATAGRAM_BYTES);
        arf.push_back(i);
        tx_recv++;
        if (remind) { write_clock(); clock_inds++; remind = false; }                              // (:785-789)
      }
    }
    if (!arf.empty()) CHK(trxsig_trxgroup_add_bursts(grp, dgs.data(), arf.data(), (int)arf.size()), "trxsig_trxgroup_add_bursts");
    // ---- driveTransmitFIFO: everything due with the radio clock at (fn, 0) ----
    for (;;) {
      int pfn = 0, ptn = 0;
      const int n = trxsig_txclock_advance(&txc, fn, 0, &pending_underrun, 64, &pfn, &ptn);
      // (the loopback ring holds R frames: a latency the controller has grown beyond R - 2 frames would overwrite frames still on the air)
      if (txc.latency_fn > air.R - 2) { txc.latency_fn = air.R - 2; txc.latency_tn = 0; }
      if (n <= 0) break;
      const uint8_t *d_bits = nullptr, *d_fq = nullptr;
      const float *d_gain = nullptr;
      CHK(trxsig_trxgroup_push(grp, pfn, ptn, n, &d_bits, &d_gain, &d_fq), "trxsig_trxgroup_push");
      if (dbg_arfcn >= 0) {                                 // --debug-arfcn A: which of A's timeslots left the queue (stderr)
        std::vector<uint8_t> fq((size_t)n);
        (void)hipMemcpy(fq.data(), d_fq + (size_t)dbg_arfcn * n, (size_t)n, hipMemcpyDeviceToHost);
        for (int t = 0; t < n; t++)
          if (fq[(size_t)t]) std::fprintf(stderr, "tx arfcn %d radio fn %d: slot %d of push (%d,%d)+%d from the queue\n", dbg_arfcn, fn, t, pfn, ptn, n);
      }
      CHK(air.transmit(d_bits, d_gain, pfn, ptn, n), "loopback transmit");
      if (n < 64) break;
    }
    // ---- driveReceiveFIFO: the frame that is on the air ----
    trxsig_trxgroup_result res;
    CHK(trxsig_trxgroup_pull(grp, air.frame(fn), (int64_t)N * air.cell, air.cell, 0, fn, 0, 8, &res), "trxsig_trxgroup_pull");
    CHK(trxsig_trxgroup_collect(grp, valid.data(), soft.data(), rssi.data(), timing.data(), nullptr), "trxsig_trxgroup_collect");
    for (int tn = 0; tn < 8; tn++)
      for (int i = 0; i < N; i++) {
        const size_t c = (size_t)tn * N + i;
        if (!valid[c]) continue;
        if (i == dbg_arfcn) std::fprintf(stderr, "rx arfcn %d fn %d tn %d rssi %d toa %d\n", i, fn, tn, rssi[c], timing[c]);
        uint8_t out[TRXSIG_RX_DATAGRAM_BYTES];
        trxsig_trx_encode_rx_datagram(tn, fn, rssi[c], timing[c], soft.data() + c * 148, 148, out);
        send_to(dat[(size_t)i], port + 102 + 2 * i, out, sizeof out);
        rx_sent++;
      }
    const double busy = std::chrono::duration<double, std::micro>(clock_t_::now() - t0).count();
    busy_sum += busy; busy_max = std::max(busy_max, busy);
  }
  int arfcns_dropped = 0;                                   // (a transmit queue holds 256 bursts per ARFCN; what did not fit was dropped and marked)
  for (int i = 0; i < N; i++) { int d = 0; if (trxsig_trxgroup_tx_queue_size(grp, i, &d) >= 0 && d) arfcns_dropped++; }
  if (arfcns_dropped) std::printf("WARNING: %d ARFCN(s) dropped transmit bursts (queue of 256 per ARFCN full)\n", arfcns_dropped);
  std::printf("frames %ld  arfcns %d  tx bursts received %ld  rx bursts sent %ld  malformed %ld  clock indications %ld  under-runs %ld  "
              "transmit latency %d:%d  service time per frame avg %.1f us max %.1f us (frame = %d us)\n",
              k, N, tx_recv, rx_sent, bad_len, clock_inds, underruns, txc.latency_fn, txc.latency_tn, k ? busy_sum / k : 0.0, busy_max, frame_us);
  trxsig_trxgroup_destroy(grp);
  (void)hipFree(air.d_ring); (void)hipFree(air.d_meta); (void)hipFree(air.d_scale);
  trxsig_destroy(ctx);
  close(clk);
  for (int i = 0; i < N; i++) { close(ctl[(size_t)i]); close(dat[(size_t)i]); }
  return 0;
}
