// trxsig_ctx.h -- internal: what the other host translation units of the library (trxsig_frontend.cpp, trxsig_trxgroup.cpp)
// may ask of a context (trxsig_api.cpp owns struct trxsig_ctx).
#pragma once
#include "trxsig.h"
#include "trxsig_launch.h"

int trx_ctx_fail(trxsig_ctx *c, int code, const char *what, hipError_t e);

// Small tables a call makes on the host and a kernel of the same call reads: pinned staging blocks taken in turn, each free again
// when the upload that read it has run.  (A PAGEABLE source makes hipMemcpyAsync wait until the stream has drained -- the host then
// cannot run ahead of the device at all: ~0.5 ms a call with the device busy, profiles/r05_group_tx_bench.txt.)
struct TrxPinRing {
  static constexpr int kSlots = 4;
  void *p[kSlots] = {nullptr, nullptr, nullptr, nullptr};
  size_t cap[kSlots] = {0, 0, 0, 0};
  hipEvent_t ev[kSlots] = {nullptr, nullptr, nullptr, nullptr};
  bool armed[kSlots] = {false, false, false, false};
  unsigned turn = 0;
  // the next block, at least `bytes` long (waits for the upload that used it four calls ago, if that has not run yet)
  hipError_t take(size_t bytes, void **out, int *slot) {
    const int k = (int)(turn++ % kSlots);
    hipError_t e = hipSuccess;
    if (!ev[k] && (e = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming)) != hipSuccess) return e;
    if (armed[k]) { if ((e = hipEventSynchronize(ev[k])) != hipSuccess) return e; armed[k] = false; }
    if (bytes > cap[k]) {
      void *q = nullptr;
      const size_t want = bytes + bytes / 4 + 256;
      if ((e = hipHostMalloc(&q, want, hipHostMallocDefault)) != hipSuccess) return e;
      if (p[k]) (void)hipHostFree(p[k]);
      p[k] = q; cap[k] = want;
    }
    *out = p[k]; *slot = k;
    return hipSuccess;
  }
  // hipMemcpyAsync(dst, block, bytes) on st, and the block's event behind it
  hipError_t upload(int slot, void *dst, size_t bytes, hipStream_t st) {
    hipError_t e = hipMemcpyAsync(dst, p[slot], bytes, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return e;
    if ((e = hipEventRecord(ev[slot], st)) != hipSuccess) return e;
    armed[slot] = true;
    return hipSuccess;
  }
  void release() {
    for (int k = 0; k < kSlots; k++) {
      if (ev[k]) { if (armed[k]) (void)hipEventSynchronize(ev[k]); (void)hipEventDestroy(ev[k]); ev[k] = nullptr; }
      if (p[k]) { (void)hipHostFree(p[k]); p[k] = nullptr; }
      cap[k] = 0; armed[k] = false;
    }
  }
};
TrxProfiler *trx_ctx_profiler(trxsig_ctx *c);
// the normal-burst leg on bursts computed from the raw int16 stream (trxsig_rxfe_push_detect_demod_normal)
// (on != nullptr: launched on that stream instead of the context's; the caller orders it against the context's stream)
int trx_ctx_rx_normal(trxsig_ctx *c, const TrxRxGen &gen, int B, int tsc, float detect_thresh, float energy_thresh, uint8_t *d_flags,
                      trxsig_c32 *d_amp, float *d_toa, float *d_avgpwr, float *d_soft, uint8_t *d_hard, int nsoft, int soft_stride,
                      hipStream_t on = nullptr);
// ... detectRACHBurst (d_len[b] = the burst's length as the front end cuts it) and demodulateBurst with caller-supplied
// amplitude / TOA for the bursts whose d_enable[b] != 0, on such bursts
// (own_records != 0: the detect -> peak records go to a scratch of the call's own, so that it may run beside trx_ctx_rx_normal)
int trx_ctx_rx_rach(trxsig_ctx *c, const TrxRxGen &gen, const int32_t *d_len, int B, float detect_thresh, float energy_thresh,
                    uint8_t *d_flags, trxsig_c32 *d_amp, float *d_toa, float *d_avgpwr, int own_records = 0);
int trx_ctx_rx_demod(trxsig_ctx *c, const TrxRxGen &gen, int B, const trxsig_c32 *d_amp, const float *d_toa, const uint8_t *d_enable,
                     int need_mask, float *d_soft, int nsoft, int soft_stride);
int trx_ctx_demod_masked(trxsig_ctx *c, const trxsig_c32 *d_samples, const int32_t *d_offset, const int32_t *d_length, int B,
                         const trxsig_c32 *d_amp, const float *d_toa, const uint8_t *d_enable, int need_mask, float *d_soft, int nsoft,
                         int soft_stride);
// the receive front end's side of a fused push (trxsig_frontend.cpp): which bursts this push completes and how the kernels
// find their samples (begin), and the window / clock bookkeeping once the kernels are enqueued (end)
struct trxsig_rxfe;
struct TrxRxfePush { TrxRxGen gen; int nb, tn0, n_streams; };
int trx_rxfe_fused_begin(trxsig_rxfe *fe, const int16_t *d_iq, int n_chunks, TrxRxfePush *out);
int trx_rxfe_fused_end(trxsig_rxfe *fe, const int16_t *d_iq, int n_chunks, const TrxRxfePush &p);
// an object that lives on a context keeps it alive: trxsig_destroy on a context with such objects takes effect when the last is gone
void trx_ctx_retain(trxsig_ctx *c);
void trx_ctx_release(trxsig_ctx *c);
struct trxsig_txbe;
trxsig_ctx *trx_txbe_context(const trxsig_txbe *be);         // the context a transmit back end was created on (trxsig_frontend.cpp)
trxsig_ctx *trx_rxfe_ctx(trxsig_rxfe *fe);
int trx_rxfe_streams(const trxsig_rxfe *fe);
int trx_rxfe_next_tn(const trxsig_rxfe *fe);
// Transceiver group (trxsig_trxgroup.cpp), equalising TSC leg (sps = 1):
//   estimate: analyzeTrafficBurst(requestChannel) + scaleVector(chan, 1/amp) + designDFE(chan, d_snr[b], 7) for the bursts with
//     d_enable[b] != 0 only (Transceiver.cpp:341-349); nothing is written for the others.  Detection threshold 3.0 (:331).
//   equalize: scaleVector(burst, 1/amp) + equalizeBurst(burst, d_toa_eq[b], w, b) (:391-396) for the bursts whose d_gate has
//     TRXSIG_F_DETECT, burst b with the taps at entry d_tap_ix[b] of the tap table.
int trx_ctx_group_estimate(trxsig_ctx *c, const trxsig_c32 *d_samples, const int32_t *d_offset, const int32_t *d_length, int B, int tsc,
                           const uint8_t *d_enable, const float *d_snr, uint8_t *d_flags, trxsig_c32 *d_amp, float *d_toa,
                           float *d_toa_eq, float *d_chan_off, trxsig_c32 *d_w, trxsig_c32 *d_b,
                           int32_t *d_listed = nullptr /* the marked bursts already listed: their count, then their indices (any order) */);
int trx_ctx_group_equalize(trxsig_ctx *c, const trxsig_c32 *d_samples, const int32_t *d_offset, const int32_t *d_length, int B,
                           const trxsig_c32 *d_amp, const float *d_toa_eq, const uint8_t *d_gate, const trxsig_c32 *d_w_tab,
                           const trxsig_c32 *d_b_tab, const int32_t *d_tap_ix, float *d_soft, int nsoft, int soft_stride);
