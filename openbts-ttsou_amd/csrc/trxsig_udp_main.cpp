// trxsig_udp_main.cpp -- a socket loop around the Transceiver object (include/trxsig_transceiver.h) that speaks
// the TRXManager UDP contract (TRXManager/README.TRXManager; Transceiver/Transceiver.cpp:40-47, 439-793):
//   base port B: clock indications  B   -> peer B+100   "IND CLOCK <fn>"
//                control            B+1 <-> peer B+101  "CMD ..." / "RSP ..."
//                data               B+2 <-> peer B+102  154-byte transmit bursts in, 158-byte receive bursts out
// There is no radio in this image, so the samples take the software-loopback route of the reference's
// SWLOOPBACK build: what pushRadioVector hands to the transmit FIFO for slot (fn, tn) comes back, scaled to the
// radio's full-scale range (x13500, RadioInterface::pushBuffer, radioInterface.cpp:149), as the receive burst of
// the same slot.  One thread, one virtual radio clock advancing a slot at a time at --slot-us microseconds.
//
//   trxsig_transceiver_udp [--port 5700] [--sps 1] [--frames 400] [--slot-us 200] [--device 0]
#include <arpa/inet.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "trxsig_transceiver.h"

namespace {

int open_udp(int local_port) {
  const int fd = socket(AF_INET, SOCK_DGRAM, 0);
  if (fd < 0) return -1;
  sockaddr_in a{};
  a.sin_family = AF_INET;
  a.sin_port = htons((uint16_t)local_port);
  a.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
  if (bind(fd, (sockaddr *)&a, sizeof a) != 0) { close(fd); return -1; }
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

}  // namespace

int main(int argc, char **argv) {
  int port = 5700, sps = 1, frames = 400, slot_us = 200, device = 0;
  for (int i = 1; i + 1 < argc; i += 2) {
    if (!std::strcmp(argv[i], "--port")) port = std::atoi(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--sps")) sps = std::atoi(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--frames")) frames = std::atoi(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--slot-us")) slot_us = std::atoi(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--device")) device = std::atoi(argv[i + 1]);
  }
  const int clk = open_udp(port), ctl = open_udp(port + 1), dat = open_udp(port + 2);
  if (clk < 0 || ctl < 0 || dat < 0) { std::fprintf(stderr, "cannot bind UDP ports %d..%d\n", port, port + 2); return 2; }
  trxsig_trx *trx = nullptr;
  const int start_fn = 2;                                   // runTransceiver.cpp starts the clock at GSM::Time(2,0)
  if (trxsig_trx_create(&trx, device, sps, start_fn, 0) != TRXSIG_OK) { std::fprintf(stderr, "no transceiver (GPU?)\n"); return 3; }

  auto write_clock = [&](int fn) {                          // writeClockInterface (:779-793): "IND CLOCK <deadline FN + 20>"
    char msg[50];
    const int n = std::snprintf(msg, sizeof msg, "IND CLOCK %llu", (unsigned long long)(fn + 20));
    send_to(clk, port + 100, msg, (size_t)n + 1);
  };
  std::vector<trxsig_c32> burst((size_t)157 * sps);
  std::vector<float> soft(160);
  long rx_sent = 0, tx_recv = 0;
  int last_clock = -1000;
  write_clock(start_fn);
  for (int fn = start_fn; fn < start_fn + frames; fn++) {
    for (int tn = 0; tn < 8; tn++) {
      char cbuf[128];
      for (long n; (n = recv_nb(ctl, cbuf, sizeof cbuf - 1)) > 0;) {           // driveControl
        cbuf[n] = 0;
        char rsp[128];
        const int rn = trxsig_trx_control(trx, cbuf, rsp, sizeof rsp);
        write_clock(fn);                                                      // (:463)
        if (rn > 0) send_to(ctl, port + 101, rsp, (size_t)rn + 1);
      }
      uint8_t dbuf[256];
      for (long n; (n = recv_nb(dat, dbuf, sizeof dbuf)) > 0;) {              // driveTransmitPriorityQueue
        int ttn, tfn, rssi;
        uint8_t bits[148];
        if (trxsig_trx_decode_tx_datagram(dbuf, (int)n, &ttn, &tfn, &rssi, bits) == TRXSIG_OK && ttn >= 0 && ttn < 8) {
          trxsig_trx_add_radio_vector(trx, bits, rssi, ttn, tfn);
          tx_recv++;
        }
      }
      if (fn - last_clock > 216) { write_clock(fn); last_clock = fn; }        // periodic clock update (:617-618)
      int n = 0, fq = 0;
      if (trxsig_trx_push_radio_vector(trx, tn, fn, burst.data(), &n, &fq) != TRXSIG_OK) return 4;
      for (int i = 0; i < n; i++) { burst[i].re *= 13500.0f; burst[i].im *= 13500.0f; }   // loopback at radio scale
      int ns = 0, rssi = 0, toa = 0;
      const int got = trxsig_trx_pull_radio_vector(trx, burst.data(), n, tn, fn, soft.data(), &ns, &rssi, &toa);
      if (got < 0) { std::fprintf(stderr, "pullRadioVector: %s\n", trxsig_trx_last_error(trx)); return 5; }
      if (got == 1) {                                                         // driveReceiveFIFO
        uint8_t out[TRXSIG_RX_DATAGRAM_BYTES];
        trxsig_trx_encode_rx_datagram(tn, fn, rssi, toa, soft.data(), ns, out);
        send_to(dat, port + 102, out, sizeof out);
        rx_sent++;
      }
      if (slot_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(slot_us));
    }
  }
  std::printf("frames %d  tx bursts received %ld  rx bursts sent %ld  final energy threshold %.3f\n", frames, tx_recv, rx_sent,
              trxsig_trx_energy_threshold(trx));
  trxsig_trx_destroy(trx);
  close(clk); close(ctl); close(dat);
  return 0;
}
